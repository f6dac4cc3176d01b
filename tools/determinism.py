#!/usr/bin/env python3
"""Run-to-run determinism of the demo pair on one stream: N repetitions, outputs compared bit for bit (a kernel-internal race would
show up here; the sizes cover the 128-row and the 64-row plane tiles and the exact-f32 kernels)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mocha_sigasia2023_amd import Generator, synthetic, weights
dev = torch.device("cuda:0")
sd = weights.synthetic_state_dict(1777, 1.0, "mixamo")
model = Generator(layout="mixamo", device=dev).load_state_dict(sd).eval()
mean, std = synthetic.cnt_norm(7)
for W in (585, 128, 40):
    src = torch.from_numpy(synthetic.pose_windows(1777, W, 22)).to(dev); cha = torch.from_numpy(synthetic.pose_windows(4242, W, 22)).to(dev)
    Y0, i0 = model.characterize_pair(src, cha, mean, std, return_index=True)
    diffs = 0
    for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 30):
        Y, i = model.characterize_pair(src, cha, mean, std, return_index=True)
        diffs += int(not (torch.equal(Y, Y0) and torch.equal(i, i0)))
    print(f"{W} + {W} windows: {diffs} repetitions differ from the first")
