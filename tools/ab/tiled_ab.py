#!/usr/bin/env python3
"""A/B of the coarse pass's tiled bank image (option "bank_tiled"): 128 queries x 4 096 bf16 rows stand-alone (repeated calls: the bank stays in
the Infinity Cache) and inside characterize(128 windows) (cold: other kernels run between two passes); also 1 024 x 4 096 and a ragged bank."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mocha_sigasia2023_amd import ContextBank, Generator, synthetic, synthetic_state_dict
dev = torch.device("cuda:0")
model = Generator(layout="mixamo", device=dev).load_state_dict(synthetic_state_dict(1777, 1.0, "mixamo")).eval()
g = torch.Generator(device=dev); g.manual_seed(16384)
big = torch.randn((4096, 23040), device=dev, generator=g)
m_, s_ = synthetic.cnt_norm(7); mean, std = torch.from_numpy(m_).to(dev), torch.from_numpy(s_).to(dev)
for Q, N in ((128, 4096), (1024, 4096), (128, 4001)):
    q = torch.randn((Q, 23040), device=dev, generator=g)
    X = torch.from_numpy(synthetic.pose_windows(123, Q, 22)).to(dev)
    ref = None
    lw = 0                                                   # (the loader-wave instance is parked: tools/experiments/match_gemm_loader_waves_r04.hip.txt)
    for tiled, npl in ((0, 1), (1, 1), (1, 2), (0, 1), (1, 1), (1, 2)):
        model.set_option("bank_tiled", tiled); model.set_option("match_planes", npl)
        bank = ContextBank(model, big[:N], big[:N].view(N, 90, 256), bf16=True)
        for _ in range(3): d, i = bank.query(q)
        model.profile_start()
        for _ in range(10): d, i = bank.query(q)
        k = model.profile_stop()["kernels"]
        if ref is None: ref = i.clone()
        for _ in range(3): Y, ic = bank.characterize(X, mean, std, return_index=True)
        model.profile_start()
        for _ in range(10): Y, ic = bank.characterize(X, mean, std, return_index=True)
        sites = model.profile_stop()["sites"]
        inpath = {s.split("|")[1].replace("mocha_", ""): v["ms"] / v["launches"] * 1e3 for s, v in sites.items() if s.startswith("match.")}
        print(f"Q={Q:5d} N={N:5d} tiled={tiled} loader_waves={lw} planes={npl}: stand-alone " + "  ".join(f"{a.replace('mocha_', '')} {v['ms'] / v['launches'] * 1e3:6.1f}" for a, v in sorted(k.items())) +
              "   | in characterize " + "  ".join(f"{a} {b:6.1f}" for a, b in sorted(inpath.items())) + f"   same indices: {bool(torch.equal(i, ref))}", flush=True)
model.set_option("bank_tiled", 0); model.set_option("match_planes", 1)
