#!/usr/bin/env python3
"""A/B of the decoder projection folding (mocha_set_option fold_decoder): error against the oracle and step time."""
import os
import sys, time, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mocha_sigasia2023_amd import Generator, ContextBank, synthetic, synthetic_state_dict
from oracle import mocha_oracle as O
dev = torch.device('cuda:0')
for gain in (1.0, 2.0):
    sd = synthetic_state_dict(1777, gain)
    model = Generator(device=dev).load_state_dict(sd).eval()
    src = torch.from_numpy(synthetic.pose_windows(1, 16)).to(dev); cha = torch.from_numpy(synthetic.pose_windows(2, 16)).to(dev)
    with torch.no_grad():
        Yo = O.generator_forward(O.to_torch_state(sd), src.cpu(), cha.cpu())
    for fold in (0, 1):
        model.set_option("fold_decoder", fold)
        Y = model(src, cha).cpu()
        print(f"gain {gain} fold {fold}: max abs err {float((Y-Yo).abs().max()):.3e}  (|Y| max {float(Yo.abs().max()):.2f})")
model = Generator(device=dev).load_state_dict(synthetic_state_dict(1777, 1.0)).eval()
W = 585
src = torch.from_numpy(synthetic.pose_windows(1, W)).to(dev); cha = torch.from_numpy(synthetic.pose_windows(2, W)).to(dev)
m_, s_ = synthetic.cnt_norm(7); mean, std = torch.from_numpy(m_).to(dev), torch.from_numpy(s_).to(dev)
def step():
    enc_c, cnt_c, nm_c = model.encode(cha, mean, std)
    return ContextBank(model, nm_c, enc_c).characterize(src, mean, std)
for fold in (0, 1, 0, 1):
    model.set_option("fold_decoder", fold)
    for _ in range(3): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): step()
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 10 * 1e3
    print(f"fold {fold}: {ms:.3f} ms/step  {W/ms*1e3:.0f} frames/s")
