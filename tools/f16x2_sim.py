#!/usr/bin/env python3
"""CPU simulation (build container, no GPU): what a TWO-plane fp16 engine with THREE matrix-pipe passes per product would do to a GEMM's
accuracy, beside the engines the library has.  Not built - DESIGN.md section 8 weighs it.

  x = x0 + x1,  x0 = fp16(s x), x1 = fp16(s x - x0)      (round to nearest even; 22 significant bits; s = a power of two per tensor / row)
  a b ~ a0 b0 + a0 b1 + a1 b0                            (the dropped a1 b1 is 2^-22 |ab|, like the planes' own truncation)

Every fp16 x fp16 product is exact in fp32; a v_mfma_f32_32x32x16_f16 adds 16 of them per instruction into an fp32 accumulator.  The
simulation rounds once per instruction (the sums inside one instruction in float64), i.e. it models the accumulator chain: what differs
between the engines is (a) the number of accumulator roundings per output - K / 2 for the exact-f32 pipe (v_mfma_f32_32x32x2_f32),
6 K / 16 for the three bf16 planes, 3 K / 16 for the two fp16 planes - and (b) the planes' truncation (none / <= 2^-24 / <= 2^-22 per product).
Output: rms and max error against the float64 product in units of the output's rms, and the worst ROW's rms error in units of that row's
rms (what a per-tensor activation scale costs rows far below the tensor's largest: fp16 keeps 22 bits over 2^18 only)."""
import sys
import torch

torch.manual_seed(0)


def planes(x, n, dtype, scale=1.0):
    out, r = [], (x * scale).clone()
    for _ in range(n):
        p = r.to(dtype).to(torch.float32)
        out.append(p)
        r = r - p
    return out


def chain(terms, K, depth):
    acc = torch.zeros(terms[0][0].shape[0], terms[0][1].shape[0], dtype=torch.float32)
    for k0 in range(0, K, depth):
        for A, B in terms:
            acc = acc + (A[:, k0:k0 + depth].double() @ B[:, k0:k0 + depth].double().T).float()
    return acc


def engines(x, w, K):
    ref = x.double() @ w.double().T
    out = {}
    out["exact-f32 pipe (depth 2)"] = chain([(x, w)], K, 2).double()
    xb, wb = planes(x, 3, torch.bfloat16), planes(w, 3, torch.bfloat16)
    out["bf16 x 3 planes, 6 passes"] = chain([(xb[i], wb[j]) for i, j in ((2, 0), (1, 1), (0, 2), (1, 0), (0, 1), (0, 0))], K, 16).double()
    # per-tensor power-of-two scales: the largest element lands just below 2^15
    sx = 2.0 ** (14 - int(torch.floor(torch.log2(x.abs().max())).item()))
    sw = 2.0 ** (14 - torch.floor(torch.log2(w.abs().amax(1, keepdim=True))))          # weights: per output row, exact and offline
    xh, wh = planes(x, 2, torch.float16, sx), planes(w, 2, torch.float16, sw)
    t4 = [(xh[1], wh[1]), (xh[1], wh[0]), (xh[0], wh[1]), (xh[0], wh[0])]
    out["fp16 x 2 planes, 4 passes"] = chain(t4, K, 16).double() / (sx * sw.T.double())
    out["fp16 x 2 planes, 3 passes"] = chain(t4[1:], K, 16).double() / (sx * sw.T.double())
    out["torch.mm fp32 (CPU BLAS)"] = (x @ w.T).double()
    scale = ref.pow(2).mean().sqrt()
    row = ref.pow(2).mean(1).sqrt()                                          # per output row: the worst row's error in units of ITS rms
    return {k: (float((v - ref).pow(2).mean().sqrt() / scale), float((v - ref).abs().max() / scale),
                float(((v - ref).pow(2).mean(1).sqrt() / row).max())) for k, v in out.items()}


def case(name, x, w):
    K = x.shape[1]
    r = engines(x, w, K)
    print(f"{name}  (M = {x.shape[0]}, N = {w.shape[0]}, K = {K})")
    for k, (rms, mx, wr) in r.items():
        print(f"    {k:28s} rms {rms:.2e}   max {mx:.2e}   worst row (rms error / that row's rms) {wr:.2e}")


M, N = 256, 256
for K in (256, 512, 1024):
    case("white noise", torch.randn(M, K), torch.randn(N, K) / K ** 0.5)
K = 512
x = torch.randn(M, K); x[:, ::7] *= 40.0                                    # 40 sigma outlier channels
case("outlier channels (x 40)", x, torch.randn(N, K) / K ** 0.5)
x = torch.randn(M, K) * torch.logspace(-4, 2, M).unsqueeze(1)                # rows spread over six decades (per-TENSOR scale)
case("rows over six decades", x, torch.randn(N, K) / K ** 0.5)
x = torch.randn(M, K) * torch.logspace(-2, 2, M).unsqueeze(1)
case("rows over four decades", x, torch.randn(N, K) / K ** 0.5)
x = torch.randn(M, 1) + 1e-3 * torch.randn(M, K)                             # nearly constant rows against zero-sum weights: cancellation
w = torch.randn(N, K); w -= w.mean(1, keepdim=True)
case("cancellation (constant rows, zero-sum weights)", x, w / K ** 0.5)
x = torch.nn.functional.gelu(torch.randn(M, K) * 3)                          # post-GELU activations (half of them ~0)
case("post-GELU", x, torch.randn(N, K) / K ** 0.5)
