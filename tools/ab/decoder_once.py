#!/usr/bin/env python3
"""A few decoder calls at B windows with the key / value image attention on or off (for rocprofv3 passes): decoder_once.py <B> <kv 0|1>"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mocha_sigasia2023_amd import Generator, synthetic, synthetic_state_dict
B, kv = int(sys.argv[1]), int(sys.argv[2])
dev = torch.device("cuda:0")
model = Generator(device=dev).load_state_dict(synthetic_state_dict(1777, 1.0)).eval()
model.set_option("attention_kv", kv)
tok = torch.from_numpy(synthetic.token_features(5, B)).to(dev); cha = torch.from_numpy(synthetic.token_features(6, B)).to(dev)
for _ in range(6):
    model.decoder(tok, cha)
torch.cuda.synchronize()
