#!/usr/bin/env python3
"""Per-site times of ContextBank.characterize at the per-GPU share of BASELINE configs[3] (128 windows, 4 096-entry bf16 bank) and at 256 / 512."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mocha_sigasia2023_amd import ContextBank, Generator, synthetic, synthetic_state_dict
dev = torch.device("cuda:0")
V = 22
model = Generator(layout="mixamo", device=dev).load_state_dict(synthetic_state_dict(1777, 1.0, "mixamo")).eval()
for kv in filter(None, os.environ.get("MOCHA_OPTS", "").split(",")):           # e.g. MOCHA_OPTS=gemm_tile64_below=0,match_pass=0
    k, v = kv.split("="); model.set_option(k, int(v)); print(f"option {k} = {v}")
g = torch.Generator(device=dev); g.manual_seed(2)
nm = torch.randn((4096, 23040), device=dev, generator=g); enc = torch.randn((4096, 90, 256), device=dev, generator=g)
bank = ContextBank(model, nm, enc, bf16=True)
m_, s_ = synthetic.cnt_norm(7); mean, std = torch.from_numpy(m_).to(dev), torch.from_numpy(s_).to(dev)
for W in (int(a) for a in (sys.argv[1:] or ["128", "256", "512"])):
    X = torch.from_numpy(synthetic.pose_windows(1, W, V)).to(dev)
    for _ in range(5): bank.characterize(X, mean, std)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50): bank.characterize(X, mean, std)
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 50 * 1e3
    model.profile_start()
    for _ in range(5): bank.characterize(X, mean, std)
    p = model.profile_stop()["sites"]
    tot = sum(v["ms"] for v in p.values()) / 5
    print(f"--- {W} windows: {ms:.3f} ms per step = {W / ms:.1f} k frames/s; kernels sum to {tot:.3f} ms in {sum(v['launches'] for v in p.values()) // 5} launches")
    for s, v in sorted(p.items(), key=lambda kv: -kv[1]["ms"])[:24]:
        print(f"   {s:60s} {v['launches'] // 5:2d} x {v['ms'] / v['launches'] * 1e3:7.1f} us   {v['flops'] / max(v['ms'], 1e-9) / 1e9:7.1f} TFLOP/s  {v['bytes'] / max(v['ms'], 1e-9) / 1e6:7.0f} GB/s")
