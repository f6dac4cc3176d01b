#!/usr/bin/env python3
"""The many-query bf16 matcher AS mocha_characterize RUNS IT (128 / 256 windows against the 4 096-entry bf16 bank): per-kernel HIP-event times of
the match.* call sites under each coarse-pass variant, alternated twice on one box.  Inside the step the caches hold the other kernels'
lines (some dirty), the bank has been evicted since the previous step: this, not the stand-alone repeat, is the state that counts.
  match_pass 0 = round 4's LDS-DMA kernel (match_nt 1: non-temporal bank loads); 1 = mocha_match_pass256 on the row-major bank;
  2 = the same from the operand-order image (variant 16: non-temporal); match_planes 2 = two query planes + select2."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mocha_sigasia2023_amd import ContextBank, Generator, synthetic, synthetic_state_dict
dev = torch.device("cuda:0")
V = 22
model = Generator(layout="mixamo", device=dev).load_state_dict(synthetic_state_dict(1777, 1.0, "mixamo")).eval()
g = torch.Generator(device=dev); g.manual_seed(2)
nm = torch.randn((4096, 23040), device=dev, generator=g); enc = torch.randn((4096, 90, 256), device=dev, generator=g)
m_, s_ = synthetic.cnt_norm(7); mean, std = torch.from_numpy(m_).to(dev), torch.from_numpy(s_).to(dev)
SETS = [("round-4 kernel", dict(match_pass=0, match_nt=0, match_planes=1, match_pass_variant=0, match_fold=0)),
        ("round-4 kernel, nt bank loads", dict(match_pass=0, match_nt=1, match_planes=1, match_pass_variant=0, match_fold=0)),
        ("round-4 kernel, nt, slabs FOLDED in the pass", dict(match_pass=0, match_nt=1, match_planes=1, match_pass_variant=0, match_fold=1)),
        ("round-4 kernel, nt, 2 planes, FOLDED", dict(match_pass=0, match_nt=1, match_planes=2, match_pass_variant=0, match_fold=1)),
        ("pass256 row-major", dict(match_pass=1, match_nt=0, match_planes=1, match_pass_variant=0, match_fold=0)),
        ("pass256 operand image", dict(match_pass=2, match_nt=0, match_planes=1, match_pass_variant=0)),
        ("pass256 operand image, nt", dict(match_pass=2, match_nt=0, match_planes=1, match_pass_variant=16)),
        ("round-4 kernel, 2 planes + select2", dict(match_pass=0, match_nt=0, match_planes=2, match_pass_variant=0)),
        ("round-4 kernel, 2 planes + select2, nt", dict(match_pass=0, match_nt=1, match_planes=2, match_pass_variant=0)),
        ("pass256 operand image, nt, 2 planes", dict(match_pass=2, match_nt=0, match_planes=2, match_pass_variant=16))]
for W in (int(a) for a in (sys.argv[1:] or ["128", "256"])):
    X = torch.from_numpy(synthetic.pose_windows(1, W, V)).to(dev)
    ref = None
    for rnd in range(2):
        for name, opts in SETS:
            for k, v in opts.items(): model.set_option(k, v)
            bank = ContextBank(model, nm, enc, bf16=True)
            for _ in range(4): Y, idx = bank.characterize(X, mean, std, return_index=True)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(40): bank.characterize(X, mean, std)
            torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 40 * 1e3
            model.profile_start()
            for _ in range(10): bank.characterize(X, mean, std)
            p = model.profile_stop()["sites"]
            mk = {k.split("|")[1].replace("mocha_", ""): v["ms"] / v["launches"] * 1e3 for k, v in p.items() if k.startswith("match.")}
            if ref is None: ref = idx.clone()
            print(f"{W:4d} windows  {name:46s} step {ms:6.3f} ms   match {sum(mk.values()):6.1f} us  [" + "  ".join(f"{k} {v:.1f}" for k, v in mk.items()) +
                  f"]   same idx {bool(torch.equal(idx, ref))}", flush=True)
            del bank
for k, v in dict(match_pass=0, match_nt=1, match_planes=1, match_pass_variant=0, match_fold=0).items(): model.set_option(k, v)
