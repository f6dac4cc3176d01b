#!/usr/bin/env python3
"""A/B of the persistent plane GEMM (option "gemm_persistent" = workgroups; 0 = mocha_gemm_x3): the demo step and its GEMM sites, alternating."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mocha_sigasia2023_amd import Generator, synthetic, synthetic_state_dict
dev = torch.device("cuda:0")
V, W = 22, 585
model = Generator(layout="mixamo", device=dev).load_state_dict(synthetic_state_dict(1777, 1.0, "mixamo")).eval()
src = torch.from_numpy(synthetic.pose_windows(1777, W, V)).to(dev); cha = torch.from_numpy(synthetic.pose_windows(4242, W, V)).to(dev)
m_, s_ = synthetic.cnt_norm(7); mean, std = torch.from_numpy(m_).to(dev), torch.from_numpy(s_).to(dev)
def run(n):
    for _ in range(3): model.characterize_pair(src, cha, mean, std)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): model.characterize_pair(src, cha, mean, std)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
for pers, maxn in ((0, 1 << 30), (768, 1 << 30), (768, 512), (768, 256), (0, 1 << 30), (768, 1 << 30), (768, 512), (768, 256), (0, 1 << 30), (768, 512)):
    model.set_option("gemm_persistent", pers); model.set_option("gemm_persistent_max_n", maxn)
    ms = run(40)
    model.profile_start()
    for _ in range(3): model.characterize_pair(src, cha, mean, std)
    p = model.profile_stop()
    k = {n: v for n, v in p["kernels"].items() if "gemm_x3" in n}
    g = sum(v["ms"] for v in k.values()) / 3
    sites = {s.split("|")[0]: v["ms"] / 3 * 1e3 for s, v in p["sites"].items() if "gemm_x3" in s}
    top = "  ".join(f"{a} {b:6.1f}" for a, b in sorted(sites.items(), key=lambda kv: -kv[1])[:8])
    print(f"gemm_persistent={pers:5d} max_n={maxn if maxn < 1 << 30 else 0:5d}: step {ms:6.3f} ms  {W / ms:7.1f} k frames/s   plane GEMMs {g:6.3f} ms per step   {top}", flush=True)
model.set_option("gemm_persistent", 768); model.set_option("gemm_persistent_max_n", 512)
