#!/usr/bin/env python3
"""A/B of a context option on the demo step (585 + 585 windows, 22 joints): python tools/option_ab.py fold_joint fold_decoder ..."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mocha_sigasia2023_amd import Generator, synthetic, synthetic_state_dict
dev = torch.device("cuda:0")
model = Generator(layout="mixamo", device=dev).load_state_dict(synthetic_state_dict(1777, 1.0, "mixamo")).eval()
src = torch.from_numpy(synthetic.pose_windows(1777, 585, 22)).to(dev)
cha = torch.from_numpy(synthetic.pose_windows(4242, 585, 22)).to(dev)
mean, std = synthetic.cnt_norm(7)
def run(n=20):
    for _ in range(3): model.characterize_pair(src, cha, mean, std)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): Y = model.characterize_pair(src, cha, mean, std)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, Y
base, Y0 = run()
print(f"baseline: {base:.3f} ms")
for opt in sys.argv[1:]:
    model.set_option(opt, 1)
    t, Y = run()
    print(f"{opt}=1: {t:.3f} ms   max |dY| = {float((Y - Y0).abs().max()):.2e}")
    model.set_option(opt, 0)
t, _ = run()
print(f"baseline again: {t:.3f} ms")
