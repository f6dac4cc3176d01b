"""The GEMM epilogues evaluate the exact-erf GELU of net/transformer.py:27 (nn.GELU()) as x * Phi(x) with Phi from ONE polynomial and one
exp2 (mocha_gelu, csrc/device_utils.h); the branch-free single-precision erf it replaced (mocha_erf) stays in the header as a utility.  Both
are checked here from the coefficients in the source (float32 fma emulated in float64) against float64 references: the erf to < 1 ulp, the
GELU to 1.2e-7 max(1, |x|) absolute and 3e-6 relative for x > -3 - below the error of 0.5 x (1 + erff(x / sqrt 2)) in float32."""
import re
import os

import numpy as np
from scipy.special import erf, erfc

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _coefficients():
    src = open(os.path.join(REPO, "mocha_sigasia2023_amd", "csrc", "device_utils.h")).read()
    body = src[src.index("float mocha_erf(float a)"):src.index("// The exact-erf GELU")]
    return [float(v) for v in re.findall(r"(-?\d+\.\d+(?:e-?\d+)?)f\b", body) if v not in ("1.0", "0.927734375")]


def test_branch_free_erf_is_accurate_to_one_ulp():
    c = _coefficients()
    assert len(c) == 14, c                       # 7 + log2(e) + 6 in source order
    f = np.float32
    fma = lambda a, b, d: (a.astype(np.float64) * b.astype(np.float64) + d.astype(np.float64)).astype(f)
    K = lambda v, like: np.full_like(like, v)
    x = np.linspace(-6, 6, 400001).astype(f)
    t, s = np.abs(x), (x * x).astype(f)
    r = fma(K(c[0], x), t, K(c[1], x)); u = fma(K(c[2], x), t, K(c[3], x))
    r = fma(r, s, u)
    for k in c[4:7]:
        r = fma(r, t, K(k, x))
    r = fma(r, t, -t)
    big = np.copysign((1.0 - np.exp2(r.astype(np.float64) * c[7])).astype(f), x)
    q = K(c[8], x)
    for k in c[9:14]:
        q = fma(q, s, K(k, x))
    small = fma(q, x, x)
    got = np.where(t > 0.927734375, big, small)
    ref = erf(x.astype(np.float64))
    assert np.abs(got - ref).max() < 1e-7
    ulp = np.abs(got - ref) / np.spacing(np.maximum(np.abs(ref), 1e-30).astype(f))
    assert ulp.max() < 1.0


def test_gelu_polynomial_is_float32_faithful():
    src = open(os.path.join(REPO, "mocha_sigasia2023_amd", "csrc", "device_utils.h")).read()
    body = src[src.index("float mocha_gelu(float x)"):src.index("// four at a time")]
    c = [float(v) for v in re.findall(r"(-?\d+\.\d+(?:e[-+]?\d+)?)f\b", body)]
    assert len(c) == 13 and abs(c[0] - 2 ** -0.5) < 1e-12 and c[11] == 4.3 and c[12] == 1.0, c      # 1/sqrt 2, ten coefficients (highest first), 4.3, 1
    # the packed four-at-a-time form must use the same ten coefficients
    body4 = src[src.index("f32x4_t mocha_gelu4(f32x4_t x)"):src.index("// Plane split")]
    c4 = [float(v) for v in re.findall(r"\{(-?\d+\.\d+(?:e[-+]?\d+)?)f,", body4)]
    assert c4 == c[1:11], (c4, c[1:11])
    f = np.float32
    fma = lambda a, b, d: (a.astype(np.float64) * b.astype(np.float64) + d.astype(np.float64)).astype(f)
    x = np.concatenate([np.linspace(-9, 9, 900001), np.linspace(-0.01, 0.01, 20001), [-100.0, -20.0, 20.0, 100.0, 0.0]]).astype(f)
    t = (np.abs(x) * f(c[0])).astype(f)
    q = np.full_like(x, f(c[1]))
    for k in c[2:11]:
        q = fma(q, t, np.full_like(x, f(k)))
    with np.errstate(over="ignore"):
        h = np.exp2(q.astype(np.float64)).astype(f)
    h = np.where(t > f(4.3), f(0), h)
    got = (x * np.where(x < 0, h, (f(1) - h).astype(f))).astype(f).astype(np.float64)
    x64 = x.astype(np.float64)
    ref = np.where(x64 < 0, 0.5 * x64 * erfc(-x64 / np.sqrt(2)), 0.5 * x64 * (1 + erf(x64 / np.sqrt(2))))
    err = np.abs(got - ref)
    assert (err / np.maximum(1.0, np.abs(x64))).max() < 1.2e-7
    m = (x64 > -3) & (x64 != 0)
    assert (err[m] / np.abs(ref[m])).max() < 3e-6
    # what it replaces: 0.5 x (1 + erf(x / sqrt 2)) evaluated in float32 with a correctly rounded erf is no better
    old = (f(0.5) * x * (f(1) + erf(x64 / np.sqrt(2)).astype(f)).astype(f)).astype(f).astype(np.float64)
    assert (err / np.maximum(1.0, np.abs(x64))).max() <= 1.5 * (np.abs(old - ref) / np.maximum(1.0, np.abs(x64))).max() + 2e-8
