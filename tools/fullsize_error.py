#!/usr/bin/env python3
"""max |Y - oracle| of the demo pair at its real size (585 + 585 windows, 22 joints) on either engine set: the number behind the
1e-4 assertion of tests/test_fullsize_parity.py.  The oracle runs on the CPU (about a minute)."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mocha_sigasia2023_amd import Generator, synthetic, weights
from oracle import mocha_oracle as O          # checker

W, V, layout = 585, 22, "mixamo"
dev = torch.device("cuda:0")
sd = weights.synthetic_state_dict(1777, 1.0, layout)
model = Generator(layout=layout, device=dev).load_state_dict(sd).eval()
src = synthetic.pose_windows(1777, W, V); cha = synthetic.pose_windows(4242, W, V)
mean, std = synthetic.cnt_norm(7)
T = lambda a: torch.from_numpy(a).to(dev)
outs = {}
for flag in (1, 0):
    model.set_option("gemm_bf16x3", flag); model.set_option("attention_bf16x3", flag)
    Y, idx = model.characterize_pair(T(src), T(cha), mean, std, return_index=True)
    outs[flag] = (Y.cpu().numpy(), idx.cpu().numpy().astype(np.int64))
ost = O.to_torch_state(sd)
with torch.no_grad():
    def enc_all(X):
        e, c = zip(*(O.encode(ost, torch.from_numpy(X[s:s + 32])) for s in range(0, W, 32)))
        return torch.cat(e), torch.cat(c)
    se, sc_ = enc_all(src); ce, cc = enc_all(cha)
    q64 = O.znorm(sc_.numpy(), mean, std).reshape(W, -1).astype(np.float64)
    k64 = O.znorm(cc.numpy(), mean, std).reshape(W, -1).astype(np.float64)
    io, _ = O.match_bruteforce(q64, k64)
    sel = ce[torch.from_numpy(io)]
    Yo = torch.cat([O.to_mot(ost, O.decoder(ost, se[s:s + 32], sel[s:s + 32])) for s in range(0, W, 32)]).numpy()
for flag, name in ((1, "bf16-pipe engines (default)"), (0, "exact-f32 MFMA engines")):
    Y, idx = outs[flag]
    same = idx == io
    d = np.abs(Y[same] - Yo[same])
    print(f"{name:30s} indices equal to the oracle's: {same.sum()} / {W};  max |Y - oracle| = {d.max():.3e}  rms = {np.sqrt((d**2).mean()):.3e}  (|Y| max {np.abs(Yo).max():.2f})")
print(f"between the two engine sets: max |dY| = {np.abs(outs[1][0] - outs[0][0]).max():.3e}")
