#!/usr/bin/env python3
"""Bank-scan (context matching) bandwidth: few queries against a large bank (BASELINE configs[2,4] shapes)."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mocha_sigasia2023_amd import ContextBank, Generator, synthetic_state_dict
dev = torch.device("cuda:0")
model = Generator(device=dev).load_state_dict(synthetic_state_dict(1, 1.0)).eval()
D = 23040
for N in (4096, 16384):
    g = torch.Generator(device=dev); g.manual_seed(N)
    bank_nm = torch.randn((N, D), device=dev, generator=g)
    bank_enc = bank_nm.view(N, 90, 256)
    for bf16 in (False, True):
        bank = ContextBank(model, bank_nm, bank_enc, bf16=bf16)
        for Q in (1, 4, 8, 128):
            q = torch.randn((Q, D), device=dev, generator=g)
            for _ in range(2): bank.query(q)
            torch.cuda.synchronize()
            model.profile_start()
            for _ in range(5): bank.query(q)
            p = model.profile_stop()
            k = [v for name, v in p["kernels"].items() if "match_stream" in name or "gemm" in name]
            ms = sum(v["ms"] for v in k) / 5
            by = sum(v["bytes"] for v in k) / 5
            fl = sum(v["flops"] for v in k) / 5
            print(f"N={N:6d} bank={'bf16' if bf16 else 'f32 '} Q={Q:4d}: {ms*1e3:9.1f} us  {by/ms/1e6:8.0f} GB/s ({by/ms/1e6/8000*100:5.1f}% of 8 TB/s)  {fl/ms/1e9:7.1f} TFLOP/s  kernels={list(p['kernels'])[:3]}")
