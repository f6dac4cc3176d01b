#!/usr/bin/env python3
"""A few demo steps with the plane GEMM's persistent instance on (argv[1] = workgroups) or off (0), for rocprofv3 passes."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mocha_sigasia2023_amd import Generator, synthetic, synthetic_state_dict
dev = torch.device("cuda:0")
V, W = 22, 585
model = Generator(layout="mixamo", device=dev).load_state_dict(synthetic_state_dict(1777, 1.0, "mixamo")).eval()
model.set_option("gemm_persistent", int(sys.argv[1]))
src = torch.from_numpy(synthetic.pose_windows(1777, W, V)).to(dev); cha = torch.from_numpy(synthetic.pose_windows(4242, W, V)).to(dev)
m_, s_ = synthetic.cnt_norm(7); mean, std = torch.from_numpy(m_).to(dev), torch.from_numpy(s_).to(dev)
for _ in range(6): model.characterize_pair(src, cha, mean, std)
torch.cuda.synchronize()
