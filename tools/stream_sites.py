#!/usr/bin/env python3
"""Per-call-site kernel times of ONE streamed window (B = 1) against a 16k bank (no graph, HIP-event pairs)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mocha_sigasia2023_amd import ContextBank, Generator, synthetic, synthetic_state_dict
dev = torch.device("cuda:0")
model = Generator(device=dev).load_state_dict(synthetic_state_dict(1777, 1.0)).eval()
N = 16384
g = torch.Generator(device=dev); g.manual_seed(2)
bank = ContextBank(model, torch.randn((N, 90 * 256), device=dev, generator=g), torch.randn((N, 90, 256), device=dev, generator=g))
m_, s_ = synthetic.cnt_norm(7)
src = torch.from_numpy(synthetic.pose_windows(5, 8)).to(dev)
for i in range(4): bank.characterize(src[i:i + 1], m_, s_)
torch.cuda.synchronize()
model.profile_start()
R = 20
for i in range(R): bank.characterize(src[i % 8:i % 8 + 1], m_, s_)
p = model.profile_stop()
tot = sum(v["ms"] for v in p["sites"].values()) / R
print(f"sum of kernel time {tot*1e3:.1f} us per window, {sum(v['launches'] for v in p['sites'].values())//R} launches")
for k, v in sorted(p["sites"].items(), key=lambda kv: -kv[1]["ms"]):
    print(f"{k:60s} {v['launches']//R:3d} {v['ms']/R*1e3:8.1f} us  {v['ms']/v['launches']*1e3:7.1f} us/launch")
