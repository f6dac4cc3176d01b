#!/usr/bin/env python3
"""A/B of the many-query selection kernels: mocha_match_select (select2=0) against mocha_match_select2 (with and without distances), on
128 x 4096 bf16 (configs[3] per GPU), 585 x 585 fp32 (the demo pair) and 1024 x 4096 bf16 (configs[2]); per-kernel HIP events."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mocha_sigasia2023_amd import ContextBank, Generator, synthetic_state_dict
dev = torch.device("cuda:0")
model = Generator(device=dev).load_state_dict(synthetic_state_dict(1, 1.0)).eval()
g = torch.Generator(device=dev); g.manual_seed(16384)
big = torch.randn((4096, 23040), device=dev, generator=g)
for Q, N, bf16 in ((128, 4096, True), (585, 585, False), (1024, 4096, True), (128, 4096, False)):
    q = torch.randn((Q, 23040), device=dev, generator=g)
    bank = ContextBank(model, big[:N], big[:N].view(N, 90, 256), bf16=bf16)
    ref = None
    for sel, want_d, npl in ((0, True, 1), (1, True, 1), (1, False, 1), (1, True, 2), (1, False, 2), (0, True, 1), (1, True, 1), (1, False, 1)):
        model.set_option("select2", sel); model.set_option("match_planes", npl)
        for _ in range(3): i = bank.query(q, return_distance=want_d); i = i[1] if want_d else i
        model.profile_start()
        for _ in range(10): i = bank.query(q, return_distance=want_d); i = i[1] if want_d else i
        k = model.profile_stop()["kernels"]
        if ref is None: ref = i.clone()
        same = bool(torch.equal(i, ref))
        print(f"Q={Q:5d} N={N:5d} bf16={int(bf16)} select2={sel} planes={npl} distances={int(want_d)}: " +
              "  ".join(f"{a.replace('mocha_', '')} {v['ms'] / v['launches'] * 1e3:7.1f}" for a, v in sorted(k.items())) + f"   same indices: {same}", flush=True)
model.set_option("select2", 1); model.set_option("match_planes", 1)
