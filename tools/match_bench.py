#!/usr/bin/env python3
"""Context-matching roofline on the shapes SURVEY.md §8(d) names (one streamed query x 16k bank; 128 queries x 4k bank), every kernel
of one mocha_match call included, HIP-event timed (bench.py's `match` records), plus 1024 queries x 4k (configs[2])."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from mocha_sigasia2023_amd import ContextBank, Generator, synthetic_state_dict
dev = torch.device("cuda:0")
model = Generator(device=dev).load_state_dict(synthetic_state_dict(1, 1.0)).eval()
for name, r in bench.match_records(model, dev).items():
    ks = "  ".join(f"{k.replace('mocha_', '')}={v:.1f}" for k, v in r["kernels"].items())
    print(f"{name:16s} {r['us']:8.1f} us  {r['GB/s']:7.0f} GB/s  {r['frac_of_hbm_peak'] * 100:5.1f} % of 8 TB/s  {r['TFLOP/s']:7.1f} TFLOP/s   [{ks}]")
D = 23040
g = torch.Generator(device=dev); g.manual_seed(3)
nm = torch.randn((4096, D), device=dev, generator=g)
q = torch.randn((1024, D), device=dev, generator=g)
for bf16 in (True, False):
    bank = ContextBank(model, nm, nm.view(-1, 90, 256), bf16=bf16)
    for _ in range(2): bank.query(q)
    torch.cuda.synchronize()
    model.profile_start()
    for _ in range(5): bank.query(q)
    p = model.profile_stop()
    us = sum(v["ms"] for v in p["kernels"].values()) / 5 * 1e3
    ks = "  ".join(f"{k.replace('mocha_', '')}={v['ms'] / 5 * 1e3:.1f}" for k, v in p["kernels"].items())
    print(f"q1024_x_4k_{'bf16' if bf16 else 'f32 '} {us:8.1f} us  {2.0 * 1024 * 4096 * D / us / 1e6:7.1f} TFLOP/s   [{ks}]")
