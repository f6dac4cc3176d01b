#!/usr/bin/env python3
"""A/B: to_mot's temporal conv over the x4-upsampled frames: literal (K = 320, N = 64), folded into one 3-tap launch (K = 192, N = 256, a third
of the weight blocks zero), folded into two 2-tap launches (K = 128, N = 128 each)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mocha_sigasia2023_amd import Generator, synthetic, synthetic_state_dict
dev = torch.device("cuda:0")
V = 22
model = Generator(layout="mixamo", device=dev).load_state_dict(synthetic_state_dict(1777, 1.0, "mixamo")).eval()
m_, s_ = synthetic.cnt_norm(7); mean, std = torch.from_numpy(m_).to(dev), torch.from_numpy(s_).to(dev)
for W in (585, 128, 32):
    src = torch.from_numpy(synthetic.pose_windows(1777, W, V)).to(dev); cha = torch.from_numpy(synthetic.pose_windows(4242, W, V)).to(dev)
    ref = None
    for fold, split in [(0, 256), (1, 1 << 30), (1, 1), (0, 256), (1, 1 << 30), (1, 1)]:
        model.set_option("fold_upsample", fold); model.set_option("upsample_split_min", split)
        for _ in range(3): Y = model.characterize_pair(src, cha, mean, std)
        if ref is None: ref = Y.clone()
        err = float((ref - Y).abs().max())
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(40): model.characterize_pair(src, cha, mean, std)
        torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 40 * 1e3
        model.profile_start()
        for _ in range(3): model.characterize_pair(src, cha, mean, std)
        p = model.profile_stop()["sites"]
        pick = {s.split("|")[0]: (v["ms"] / 3 * 1e3, v["launches"] // 3) for s, v in p.items() if "tcn_joint" in s or "final_proj" in s}
        print(f"windows {W:4d} fold={fold} two-launch={'yes' if split == 1 else 'no '}: step {ms:6.3f} ms  |dY| {err:.1e}  " + "  ".join(f"{a} {b[0]:6.1f} us ({b[1]})" for a, b in sorted(pick.items())), flush=True)
model.set_option("fold_upsample", 1); model.set_option("upsample_split_min", 256)
