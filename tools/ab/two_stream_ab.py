#!/usr/bin/env python3
"""Experiment: do two independent encode chains overlap usefully on two HIP streams (two contexts)?"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from mocha_sigasia2023_amd import Generator, synthetic, synthetic_state_dict
dev = torch.device("cuda:0")
sd = synthetic_state_dict(1777, 1.0)
ma = Generator(device=dev).load_state_dict(sd); mb = Generator(device=dev).load_state_dict(sd)
W = 585
xa = torch.from_numpy(synthetic.pose_windows(1, W)).to(dev); xb = torch.from_numpy(synthetic.pose_windows(2, W)).to(dev)
x2 = torch.cat([xa, xb])
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
def seq():
    ma.encode(xa); ma.encode(xb)
def batched():
    ma.encode(x2)
def conc():
    with torch.cuda.stream(sa): ma.encode(xa)
    with torch.cuda.stream(sb): mb.encode(xb)
def halves4():
    with torch.cuda.stream(sa): ma.encode(xa[:293]); ma.encode(xa[293:])
    with torch.cuda.stream(sb): mb.encode(xb[:293]); mb.encode(xb[293:])
for name, fn in (("sequential 585+585", seq), ("one batch of 1170", batched), ("two streams 585|585", conc), ("two streams, 4 halves", halves4), ("sequential 585+585", seq)):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): fn()
    torch.cuda.synchronize(); print(f"{name:28s} {(time.perf_counter()-t0)/10*1e3:7.3f} ms")
