#!/usr/bin/env python3
"""Randomised parity sweep of the HIP path against the CPU oracle: seeds x gains x layouts x batch shapes, NN branch end to end
(encode, z-score, match, gather, decode, to_mot) through mocha_characterize_pair and through the three-call path.
Prints one line per configuration and a summary; exits non-zero on any violation of the 1e-4 tolerance or an index mismatch
that is not a tie at fp32 feature precision."""
import sys, time
import numpy as np, torch
sys.path.insert(0, __file__.rsplit('/', 2)[0])
from mocha_sigasia2023_amd import ContextBank, Generator, synthetic, weights
from oracle import mocha_oracle as O

dev = torch.device("cuda:0")
bad = 0
t0 = time.time()
cfgs = [(seed, gain, layout, bs, bc) for seed in (101, 202, 303) for gain in (0.7, 1.0, 1.6)
        for layout, bs, bc in (("mocha", 5, 9), ("mixamo", 17, 6), ("mocha", 33, 40))]
for seed, gain, layout, bs, bc in cfgs:
    V = 24 if layout == "mocha" else 22
    sd = weights.synthetic_state_dict(seed, gain, layout)
    model = Generator(layout=layout, device=dev).load_state_dict(sd).eval()
    mean, std = synthetic.cnt_norm(seed % 17)
    src = synthetic.pose_windows(seed + 1, bs, V); cha = synthetic.pose_windows(seed + 2, bc, V)
    ts, tc = torch.from_numpy(src).to(dev), torch.from_numpy(cha).to(dev)
    Yp, ip = model.characterize_pair(ts, tc, mean, std, return_index=True)
    e, c, n = model.encode(tc, mean, std)
    Y3, i3 = ContextBank(model, n, e).characterize(ts, mean, std, return_index=True)
    ost = O.to_torch_state(sd)
    with torch.no_grad():
        Yo, io = O.characterize(ost, torch.from_numpy(src), torch.from_numpy(cha), mean, std)
        qs = O.znorm(O.encode(ost, torch.from_numpy(src))[1].numpy(), mean, std).reshape(bs, -1).astype(np.float64)
        ks = O.znorm(O.encode(ost, torch.from_numpy(cha))[1].numpy(), mean, std).reshape(bc, -1).astype(np.float64)
    ours = ip.cpu().numpy()
    d_ours = np.sqrt(((qs - ks[ours]) ** 2).sum(1)); d_best = np.sqrt(((qs - ks[io]) ** 2).sum(1))
    ties_ok = bool(np.all(d_ours <= d_best * (1 + 1e-6)))
    same = ours == io
    scale = max(1.0, float(Yo.abs().max()))
    err = float((Yp.cpu()[torch.from_numpy(same)] - Yo[torch.from_numpy(same)]).abs().max()) if same.any() else 0.0
    pair_vs_three = float((Yp - Y3).abs().max()) / scale
    ok = ties_ok and err < 1e-4 * scale and bool(torch.equal(ip, i3)) and pair_vs_three < 2e-6 and bool(torch.isfinite(Yp).all())
    bad += not ok
    print(f"seed {seed} gain {gain} {layout:6s} src {bs:3d} cha {bc:3d}: |Y| max {scale:6.2f}  err {err:.2e}  idx equal {int(same.sum())}/{bs}"
          f" ties_ok {ties_ok}  pair-vs-three {pair_vs_three:.1e}  {'OK' if ok else 'FAIL'}")
print(f"{len(cfgs) - bad}/{len(cfgs)} configurations OK in {time.time() - t0:.0f} s")
sys.exit(1 if bad else 0)
