#!/usr/bin/env python3
"""A/B: instance-norm kernels with a window's channels over four workgroups (the small-batch variant) at the demo step's batch sizes."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mocha_sigasia2023_amd import Generator, synthetic, synthetic_state_dict
dev = torch.device("cuda:0")
V, W = 22, 585
model = Generator(layout="mixamo", device=dev).load_state_dict(synthetic_state_dict(1777, 1.0, "mixamo")).eval()
src = torch.from_numpy(synthetic.pose_windows(1777, W, V)).to(dev); cha = torch.from_numpy(synthetic.pose_windows(4242, W, V)).to(dev)
m_, s_ = synthetic.cnt_norm(7); mean, std = torch.from_numpy(m_).to(dev), torch.from_numpy(s_).to(dev)
qd = 16
for mx in (32, 1 << 30, 32, 1 << 30):
    model.set_option("inorm_split_max", mx)
    for _ in range(3): model.characterize_pair(src, cha, mean, std)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(40): model.characterize_pair(src, cha, mean, std)
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 40 * 1e3
    model.profile_start()
    for _ in range(3): model.characterize_pair(src, cha, mean, std)
    p = model.profile_stop()["sites"]
    pick = {s.split("|")[0]: v["ms"] / v["launches"] * 1e3 for s, v in p.items() if "instnorm" in s or "adain" in s}
    print(f"inorm_split_max={mx if mx < 1 << 30 else 'all':>4} quads={qd:2d}: step {ms:6.3f} ms   " + "  ".join(f"{a} {b:6.1f}" for a, b in sorted(pick.items())), flush=True)
model.set_option("inorm_split_max", 1 << 30)
