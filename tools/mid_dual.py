#!/usr/bin/env python3
"""The per-GPU share of configs[3] (128 windows x 4 096-entry bf16 bank) with and without the two-stream overlap, and 256 windows."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mocha_sigasia2023_amd import ContextBank, Generator, synthetic, synthetic_state_dict
dev = torch.device("cuda:0")
V = 22
g = torch.Generator(device=dev); g.manual_seed(2)
nm = torch.randn((4096, 23040), device=dev, generator=g); enc = torch.randn((4096, 90, 256), device=dev, generator=g)
m_, s_ = synthetic.cnt_norm(7); mean, std = torch.from_numpy(m_).to(dev), torch.from_numpy(s_).to(dev)
for W in (int(a) for a in (sys.argv[1:] or ["128", "256"])):
    X = torch.from_numpy(synthetic.pose_windows(1, W, V)).to(dev)
    ref = None
    for dual, dmin in ((0, 128), (1, 64), (0, 128), (1, 64)):
        model = Generator(layout="mixamo", device=dev).load_state_dict(synthetic_state_dict(1777, 1.0, "mixamo")).eval()
        model.set_option("dual_stream", dual); model.set_option("dual_min", dmin)
        bank = ContextBank(model, nm, enc, bf16=True)
        for _ in range(5): Y, i = bank.characterize(X, mean, std, return_index=True)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(100): Y, i = bank.characterize(X, mean, std, return_index=True)
        torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 100 * 1e3
        if ref is None: ref = (Y.clone(), i.clone())
        print(f"{W} windows dual_stream={dual}: {ms:.3f} ms per step = {W / ms:.1f} k frames/s   same idx {bool(torch.equal(i, ref[1]))}  max |dY| {float((Y - ref[0]).abs().max()):.1e}", flush=True)
        del model, bank
