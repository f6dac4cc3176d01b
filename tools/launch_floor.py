#!/usr/bin/env python3
"""Floor of a dependent chain of tiny kernels on this GPU (stream-ordered launches, and the same chain replayed as a HIP graph):
what a streamed window's ~44 launches cost before any of them does work."""
import time, torch
dev = torch.device("cuda:0")
x = torch.zeros(64, device=dev)
def chain(n):
    for _ in range(n): x.add_(1.0)
for n in (44, 440):
    chain(n); torch.cuda.synchronize()
    t0 = time.perf_counter(); chain(n); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g): chain(n)
    g.replay(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20): g.replay()
    torch.cuda.synchronize(); dg = (time.perf_counter() - t0) / 20
    print(f"{n} dependent tiny kernels: eager {dt*1e6:.0f} us ({dt/n*1e6:.2f} us each), graph replay {dg*1e6:.0f} us ({dg/n*1e6:.2f} us each)")
