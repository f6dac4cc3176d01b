#!/usr/bin/env python3
"""Diagnostic that exposed the ranking problem of the first matcher: HIP features vs oracle features, the exact arg-min on
the HIP features vs what the matcher returns, on a synthetic clip pair whose bank rows are near-duplicates (see DESIGN.md,
Context matching)."""
import os
import sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mocha_sigasia2023_amd import ContextBank, Generator, synthetic, weights
from oracle import mocha_oracle as O
sd = weights.synthetic_state_dict(21, 1.2)
model = Generator(device='cuda:0').load_state_dict(sd).eval()
mean, std = synthetic.cnt_norm(4)
src = torch.from_numpy(synthetic.pose_windows(5, 37)).cuda(); cha = torch.from_numpy(synthetic.pose_windows(6, 53)).cuda()
enc_c, cnt_c, nm_c = model.encode(cha, mean, std)
enc_s, cnt_s, nm_s = model.encode(src, mean, std)
ost = O.to_torch_state(sd)
with torch.no_grad():
    qo = O.znorm(O.encode(ost, src.cpu())[1].numpy(), mean, std).reshape(37, -1).astype(np.float64)
    ko = O.znorm(O.encode(ost, cha.cpu())[1].numpy(), mean, std).reshape(53, -1).astype(np.float64)
qh = nm_s.cpu().numpy().reshape(37, -1).astype(np.float64); kh = nm_c.cpu().numpy().reshape(53, -1).astype(np.float64)
print('feature err q', np.abs(qh - qo).max(), 'k', np.abs(kh - ko).max(), 'max|q|', np.abs(qo).max())
Do = np.sqrt(((qo[:, None] - ko[None]) ** 2).sum(-1)); Dh = np.sqrt(((qh[:, None] - kh[None]) ** 2).sum(-1))
print('dist matrix err', np.abs(Do - Dh).max())
bank = ContextBank(model, nm_c, enc_c)
d, i = bank.query(nm_s, k=1)
i = i[:, 0].cpu().numpy(); d = d[:, 0].cpu().numpy()
io = Do.argmin(1); ih = Dh.argmin(1)
print('oracle-feature argmin vs hip-feature argmin equal:', np.array_equal(io, ih))
bad = np.nonzero(i != ih)[0]
print('matcher vs exact argmin on its own features: mismatches', bad)
for b in bad:
    print(b, 'matcher', i[b], d[b], 'exact', ih[b], Dh[b, ih[b]], 'Dh at matcher choice', Dh[b, i[b]], 'sorted', np.sort(Dh[b])[:4])
