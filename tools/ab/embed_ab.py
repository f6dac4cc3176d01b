#!/usr/bin/env python3
"""A/B: embed_front's grid cap (workgroups that each pay the per-lane plane constants) and its 3-waves-per-SIMD build, at the demo step."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mocha_sigasia2023_amd import Generator, synthetic, synthetic_state_dict
dev = torch.device("cuda:0")
V, W = 22, int(os.environ.get("W", 585))
model = Generator(layout="mixamo", device=dev).load_state_dict(synthetic_state_dict(1777, 1.0, "mixamo")).eval()
src = torch.from_numpy(synthetic.pose_windows(1777, W, V)).to(dev); cha = torch.from_numpy(synthetic.pose_windows(4242, W, V)).to(dev)
m_, s_ = synthetic.cnt_norm(7); mean, std = torch.from_numpy(m_).to(dev), torch.from_numpy(s_).to(dev)
ref = None
for var, cap in [(0, 512), (1, 512), (1, 768), (1, 384), (0, 512), (1, 512)]:
    model.set_option("embed_sums", var); model.set_option("embed_front_max_wgs", cap)
    for _ in range(3): Y = model.characterize_pair(src, cha, mean, std)
    Y = Y[0] if isinstance(Y, (tuple, list)) else Y
    if ref is None: ref = Y.clone()
    same = bool(torch.equal(ref, Y))
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(40): model.characterize_pair(src, cha, mean, std)
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 40 * 1e3
    model.profile_start()
    for _ in range(3): model.characterize_pair(src, cha, mean, std)
    p = model.profile_stop()["sites"]
    pick = {s.split("|")[0]: v["ms"] / v["launches"] * 1e3 for s, v in p.items() if "emb.front" in s or "window_sums" in s}
    print(f"embed_sums={var} cap={cap:>8}: step {ms:6.3f} ms  same={same}  " + "  ".join(f"{a} {b:6.1f}" for a, b in sorted(pick.items())), flush=True)
