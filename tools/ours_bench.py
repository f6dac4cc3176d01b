#!/usr/bin/env python3
"""Per-frame latency of the demo's CVAE ("Ours") branch (OursSession.step: condition, CVAE sample, de-normalise, decoder,
to_mot), with and without HIP-graph replay, for 1 and 8 clips in lock step."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mocha_sigasia2023_amd import CVAE, Generator, OursSession, synthetic, synthetic_state_dict
from mocha_sigasia2023_amd import weights as W
dev = torch.device("cuda:0")
model = Generator(device=dev).load_state_dict(synthetic_state_dict(1777, 1.0)).eval()
cvae = CVAE(device=dev).load_state_dict(W.synthetic_cvae_state_dict(99, 1.0)).eval()
rng = np.random.Generator(np.random.PCG64(0))
stats = [(0.1 * rng.standard_normal((90, 256))).astype(np.float32), rng.uniform(0.5, 1.5, (90, 256)).astype(np.float32),
         (0.1 * rng.standard_normal((90, 256))).astype(np.float32), rng.uniform(0.5, 1.5, (90, 256)).astype(np.float32)]
for B in (1, 8):
    enc, cnt = model.encode(torch.from_numpy(synthetic.pose_windows(3, B)).to(dev))
    for use_graph in (False, True):
        s = OursSession(model, cvae, *stats, use_graph=use_graph).reset(enc)
        for _ in range(5): s.step(enc, cnt)
        torch.cuda.synchronize()
        lat = []
        for _ in range(200):
            t0 = time.perf_counter(); y, c = s.step(enc, cnt); y[0, 0, 0, 0].item(); lat.append(time.perf_counter() - t0)
        lat = np.array(lat) * 1e3
        print(f"clips={B} graph={use_graph}: p50 {np.percentile(lat, 50):.3f} ms  p99 {np.percentile(lat, 99):.3f} ms  "
              f"{B / np.percentile(lat, 50) * 1e3:.0f} frames/s")
