#!/usr/bin/env python3
"""20 many-query matches of Q queries against an N-row bf16 bank with the selection kernel chosen on the command line (for rocprofv3 passes):
select_once.py <Q> <N> <select2 0|1>"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mocha_sigasia2023_amd import ContextBank, Generator, synthetic_state_dict
Q, N, sel = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
dev = torch.device("cuda:0")
model = Generator(device=dev).load_state_dict(synthetic_state_dict(1, 1.0)).eval()
model.set_option("select2", sel)
g = torch.Generator(device=dev); g.manual_seed(16384)
big = torch.randn((N, 23040), device=dev, generator=g)
q = torch.randn((Q, 23040), device=dev, generator=g)
bank = ContextBank(model, big, big.view(N, 90, 256), bf16=True)
for _ in range(20):
    bank.query(q)
torch.cuda.synchronize()
