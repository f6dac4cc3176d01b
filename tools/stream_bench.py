#!/usr/bin/env python3
"""BASELINE configs[4]: T=300 clip streamed window by window (285 windows) against a 16k-entry bank,
one captured HIP graph replay per window.  Reports steps/s and p50/p99 step latency."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mocha_sigasia2023_amd import ContextBank, Generator, StreamingCharacterizer, synthetic, synthetic_state_dict
dev = torch.device("cuda:0")
model = Generator(device=dev).load_state_dict(synthetic_state_dict(1777, 1.0)).eval()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
g = torch.Generator(device=dev); g.manual_seed(2)
bank_nm = torch.randn((N, 90 * 256), device=dev, generator=g)
bank_enc = torch.randn((N, 90, 256), device=dev, generator=g)
m_, s_ = synthetic.cnt_norm(7)
src = torch.from_numpy(synthetic.pose_windows(5, 285)).to(dev)
for bf16 in (False, True):
    bank = ContextBank(model, bank_nm, bank_enc, bf16=bf16)
    for use_graph in (False, True):
        sc = StreamingCharacterizer(bank, m_, s_, use_graph=use_graph)
        for i in range(10): sc.step(src[i])
        torch.cuda.synchronize()
        lat = []
        t00 = time.perf_counter()
        for i in range(285):
            t0 = time.perf_counter(); y, idx = sc.step(src[i]); idx.item(); lat.append(time.perf_counter() - t0)
        tot = time.perf_counter() - t00
        lat = np.array(lat) * 1e3
        print(f"bank={N} {'bf16' if bf16 else 'f32 '} graph={use_graph}: {285/tot:8.1f} windows/s  p50 {np.percentile(lat,50):.3f} ms  p99 {np.percentile(lat,99):.3f} ms")
