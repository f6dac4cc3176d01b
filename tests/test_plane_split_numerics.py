"""The arithmetic behind the plane engine (gemm_x3.hip, attention_x3.hip), checked on the CPU with NumPy.

Claims (csrc/gemm_x3.hip header):
  1. every finite fp32 value whose planes stay normal in bf16 is the EXACT sum of three bf16 values
        x0 = bf16(x), x1 = bf16(x - x0), x2 = bf16(x - x0 - x1)          (round to nearest even)
     or at worst leaves a residual below 2^-24 of |x| (two ulps short at the far end of a binade);
  2. a bf16 x bf16 product is exact in fp32 (8 + 8 significant bits);
  3. the six products  a0b0 + (a0b1 + a1b0) + (a0b2 + a1b1 + a2b0)  differ from the exact a*b by at most 2^-24 of |ab| -
     the rounding of ONE fp32 operation, and two orders below the error an fp32 dot product of a few hundred terms accumulates.
These are the facts that make the engine an fp32 GEMM rather than a reduced-precision one; the GPU tests
(tests/test_gemm_engines.py) then measure the end result against float64."""
import numpy as np


def bf16_rn(x):
    """Round-to-nearest-even bf16 of float32 values, returned as float32 (what v_cvt_pk_bf16_f32 computes)."""
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) >> 16
    return (u.astype(np.uint32) << 16).view(np.float32)


def planes(x):
    x = np.asarray(x, dtype=np.float32)
    x0 = bf16_rn(x)
    r1 = (x - x0).astype(np.float32)            # exact in fp32 (Sterbenz-type cancellation)
    x1 = bf16_rn(r1)
    r2 = (r1 - x1).astype(np.float32)
    x2 = bf16_rn(r2)
    return x0, x1, x2


def _samples(n, seed):
    rng = np.random.default_rng(seed)
    mant = rng.uniform(1.0, 2.0, n)
    expo = rng.integers(-60, 60, n)
    sign = rng.choice([-1.0, 1.0], n)
    return (sign * mant * np.exp2(expo)).astype(np.float32)


def test_three_planes_reconstruct_fp32():
    x = np.concatenate([_samples(400_000, 1), np.float32([0.0, 1.0, -1.0, 3.0, 1.9999999, 1.0000001, 65504.0, 1e-30, -7.0e20])])
    x0, x1, x2 = planes(x)
    rec = x0.astype(np.float64) + x1.astype(np.float64) + x2.astype(np.float64)
    err = np.abs(rec - x.astype(np.float64))
    scale = np.maximum(np.abs(x.astype(np.float64)), 1e-300)
    assert (err / scale).max() <= 2.0 ** -24          # never worse than half an fp32 ulp relative
    assert (err == 0).mean() > 0.99                   # and exact for practically every value
    # plane magnitudes: each plane is at most half a bf16 ulp of the previous one
    nz = x0 != 0
    assert (np.abs(x1[nz]) <= np.abs(x0[nz]) * 2.0 ** -8).all()
    assert (np.abs(x2[nz]) <= np.abs(x0[nz]) * 2.0 ** -16).all()


def test_bf16_products_are_exact_in_fp32():
    a = bf16_rn(_samples(200_000, 2))
    b = bf16_rn(_samples(200_000, 3))
    p32 = (a * b).astype(np.float32)                  # one fp32 multiply
    assert np.array_equal(p32.astype(np.float64), a.astype(np.float64) * b.astype(np.float64))


def test_six_products_reproduce_the_fp32_product():
    a, b = _samples(300_000, 4), _samples(300_000, 5)
    A, B = planes(a), planes(b)
    exact = a.astype(np.float64) * b.astype(np.float64)
    six = np.zeros_like(exact)
    for i, j in ((0, 2), (1, 1), (2, 0), (0, 1), (1, 0), (0, 0)):         # the kernel's order: low-order products first
        six += A[i].astype(np.float64) * B[j].astype(np.float64)
    rel = np.abs(six - exact) / np.abs(exact)
    assert rel.max() <= 2.0 ** -24
    # dropping the three second-order products would not do: that is the 2^-17 of a two-plane scheme
    three = sum(A[i].astype(np.float64) * B[j].astype(np.float64) for i, j in ((0, 1), (1, 0), (0, 0)))
    assert (np.abs(three - exact) / np.abs(exact)).max() > 2.0 ** -19


def test_dot_product_error_is_the_accumulations():
    """A K = 512 dot product: plane products accumulated in fp32 (what the MFMA does) against an fp32 FMA chain, both measured
    against float64 - the plane engine's error is not larger, because its products are exact and only the accumulation rounds."""
    rng = np.random.default_rng(6)
    K, R = 512, 2000
    a = rng.standard_normal((R, K)).astype(np.float32)
    b = rng.standard_normal((R, K)).astype(np.float32)
    ref = (a.astype(np.float64) * b.astype(np.float64)).sum(1)
    acc = np.zeros(R, dtype=np.float32)
    for k in range(K):                                 # fp32 chain: one rounding per product-and-add (fma)
        acc = (acc.astype(np.float64) + a[:, k].astype(np.float64) * b[:, k].astype(np.float64)).astype(np.float32)
    e_f32 = np.abs(acc.astype(np.float64) - ref)
    A, B = planes(a), planes(b)
    acc = np.zeros(R, dtype=np.float32)
    for k0 in range(0, K, 16):                         # per K step of 16: six MFMA passes, each modelled as a 16-term block sum added to the fp32 accumulator
        for i, j in ((0, 2), (1, 1), (2, 0), (0, 1), (1, 0), (0, 0)):
            blk = (A[i][:, k0:k0 + 16].astype(np.float64) * B[j][:, k0:k0 + 16].astype(np.float64)).sum(1)
            acc = (acc.astype(np.float64) + blk).astype(np.float32)
    e_x3 = np.abs(acc.astype(np.float64) - ref)
    assert np.sqrt((e_x3 ** 2).mean()) <= np.sqrt((e_f32 ** 2).mean())
