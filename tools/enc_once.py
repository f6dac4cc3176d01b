#!/usr/bin/env python3
"""One 1170-window encoder pass (two layers: qkv GEMM, attention, out-proj, FF1, FF2) - a small target for PMC passes."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mocha_sigasia2023_amd import Generator, synthetic, synthetic_state_dict
m = Generator(layout="mixamo", device="cuda:0").load_state_dict(synthetic_state_dict(1777, 1.0, "mixamo")).eval()
tok = torch.from_numpy(synthetic.token_features(3, 1170)).cuda()
for _ in range(3):
    m.encoder(tok)
torch.cuda.synchronize()
