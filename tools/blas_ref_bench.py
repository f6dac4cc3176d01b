#!/usr/bin/env python3
"""What the vendor SGEMM reaches on the path's GEMM shapes (context for the roofline numbers; not part of the product)."""
import time, torch
dev = torch.device("cuda:0")
shapes = [(52650, 1536, 256), (52650, 1024, 256), (52650, 256, 1024), (52650, 256, 512), (52650, 256, 768), (52650, 512, 256), (4096, 4096, 4096)]
for M, N, K in shapes:
    a = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev)
    for _ in range(3): c = a @ w.t()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): c = a @ w.t()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
    print(f"M={M} N={N} K={K}: {dt*1e6:8.1f} us  {2.0*M*N*K/dt/1e12:6.1f} TFLOP/s")
