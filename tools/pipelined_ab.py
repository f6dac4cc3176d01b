#!/usr/bin/env python3
"""The demo step serial on one context against consecutive steps alternating on two contexts / streams (bench.py's `pipelined_steps`), alternating, three rounds."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mocha_sigasia2023_amd import Generator, synthetic, synthetic_state_dict
dev = torch.device("cuda:0")
sd = synthetic_state_dict(1777, 1.0, "mixamo")
models = [Generator(layout="mixamo", device=dev).load_state_dict(sd).eval() for _ in range(2)]
streams = [torch.cuda.Stream(device=dev) for _ in range(2)]
src = torch.from_numpy(synthetic.pose_windows(1777, 585, 22)).to(dev); cha = torch.from_numpy(synthetic.pose_windows(4242, 585, 22)).to(dev)
mean, std = synthetic.cnt_norm(7)
def serial(n):
    for _ in range(n): models[0].characterize_pair(src, cha, mean, std)
def piped(n):
    for i in range(n):
        with torch.cuda.stream(streams[i % 2]): models[i % 2].characterize_pair(src, cha, mean, std)
for f in (serial, piped): f(4); torch.cuda.synchronize()
for rnd in range(3):
    for name, f in (("serial", serial), ("two contexts", piped)):
        torch.cuda.synchronize(); t0 = time.perf_counter(); f(40); torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(f"round {rnd} {name:13s}: {dt / 40 * 1e3:.3f} ms per step = {585 * 40 / dt / 1e3:.1f} k frames/s")
