#!/usr/bin/env python3
"""A/B of the twelve-wave small-batch attention (option "attention_split_max": 0 = off) on encoder + decoder calls of B windows,
alternating on one box, and on the streamed window (configs[4], fp32 bank)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mocha_sigasia2023_amd import ContextBank, Generator, StreamingCharacterizer, synthetic, synthetic_state_dict
dev = torch.device("cuda:0")
model = Generator(device=dev).load_state_dict(synthetic_state_dict(1777, 1.0)).eval()
def timed(fn, n=200):
    for _ in range(20): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6
for B in (1, 2, 4, 8, 16, 32, 48, 64, 96, 128):
    tok = torch.from_numpy(synthetic.token_features(5, B)).to(dev); cha = torch.from_numpy(synthetic.token_features(6, B)).to(dev)
    row = []
    for mx in (0, 1 << 20, 0, 1 << 20):
        model.set_option("attention_split_max", mx)
        row.append((timed(lambda: model.encoder(tok)), timed(lambda: model.decoder(tok, cha))))
    print(f"B={B:4d}  encoder off/on/off/on: " + " ".join(f"{r[0]:7.1f}" for r in row) + "   decoder: " + " ".join(f"{r[1]:7.1f}" for r in row) + " us")
g = torch.Generator(device=dev); g.manual_seed(2)
N = 16384
bank_nm = torch.randn((N, 90 * 256), device=dev, generator=g)
m_, s_ = synthetic.cnt_norm(7)
src = torch.from_numpy(synthetic.pose_windows(5, 285)).to(dev)
bank = ContextBank(model, bank_nm, bank_nm.view(N, 90, 256), bf16=False)
for mx in (0, 192, 0, 192):
    model.set_option("attention_split_max", mx)
    sc = StreamingCharacterizer(bank, m_, s_, use_graph=True)
    for i in range(10): sc.step(src[i])
    torch.cuda.synchronize(); lat = []
    for i in range(285):
        t0 = time.perf_counter(); y, idx = sc.step(src[i]); idx.item(); lat.append(time.perf_counter() - t0)
    print(f"streamed window, fp32 bank, split_max={mx}: p50 {np.percentile(np.array(lat) * 1e3, 50):.3f} ms")
