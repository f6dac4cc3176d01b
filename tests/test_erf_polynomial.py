"""The GEMM epilogues and the fused transformer tail evaluate the exact-erf GELU of net/transformer.py:27 (nn.GELU()) with a
branch-free single-precision erf (csrc/device_utils.h); this checks the polynomial it uses (same coefficients, float32 fma emulated in float64)
against a float64 erf: < 1 ulp, i.e. below the float32 resolution of the reference's own erff."""
import re
import os

import numpy as np
from scipy.special import erf

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _coefficients():
    src = open(os.path.join(REPO, "mocha_sigasia2023_amd", "csrc", "device_utils.h")).read()
    body = src[src.index("float mocha_erf(float a)"):src.index("}  // namespace mocha")]
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
