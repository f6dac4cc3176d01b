#!/usr/bin/env python3
"""Consecutive mid-size steps overlapped: two (three) contexts of one process on their own streams characterize alternate batches of W windows
against the same 4 096-entry bf16 bank (borrowed: one copy of the rows, derived data per context).  At the per-GPU shares of configs[3] a
step is a latency chain of ~37 launches that leaves most of the chip idle; independent steps fill each other's gaps (include/mocha_hip.h:
"several contexts of one process may be driven concurrently from different streams").  Prints frames/s for 1, 2 and 3 contexts and checks that
every context returns the same poses as the serial call."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mocha_sigasia2023_amd import ContextBank, Generator, synthetic, synthetic_state_dict
dev = torch.device("cuda:0")
V = 22
sd = synthetic_state_dict(1777, 1.0, "mixamo")
g = torch.Generator(device=dev); g.manual_seed(2)
nm = torch.randn((4096, 23040), device=dev, generator=g); enc = torch.randn((4096, 90, 256), device=dev, generator=g)
m_, s_ = synthetic.cnt_norm(7); mean, std = torch.from_numpy(m_).to(dev), torch.from_numpy(s_).to(dev)
NCTX = 3
models = [Generator(layout="mixamo", device=dev).load_state_dict(sd).eval() for _ in range(NCTX)]
banks = [ContextBank(m, nm, enc, bf16=True) for m in models]
streams = [torch.cuda.Stream(device=dev) for _ in range(NCTX)]
for W in (int(a) for a in (sys.argv[1:] or ["128", "256", "512"])):
    Xs = [torch.from_numpy(synthetic.pose_windows(10 + k, W, V)).to(dev) for k in range(6)]
    ref = [banks[0].characterize(x, mean, std) for x in Xs]
    torch.cuda.synchronize()
    for n in range(1, NCTX + 1):
        def run(steps):
            outs = []
            for i in range(steps):
                k = i % n
                with torch.cuda.stream(streams[k]):
                    outs.append(banks[k].characterize(Xs[i % len(Xs)], mean, std))
            return outs
        run(2 * n); torch.cuda.synchronize()
        steps = 60
        t0 = time.perf_counter(); outs = run(steps); torch.cuda.synchronize(); dt = time.perf_counter() - t0
        same = all(torch.equal(outs[i], ref[i % len(Xs)]) for i in range(len(Xs)))
        print(f"{W:4d} windows, {n} context(s): {dt / steps * 1e3:6.3f} ms per step = {W * steps / dt / 1e3:6.1f} k frames/s   same poses {same}", flush=True)
