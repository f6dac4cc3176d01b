#!/usr/bin/env python3
"""Which part of the HIP path loses accuracy on 30-50 sigma outliers (tests/test_structured_inputs.py, 'outliers in both')?  The same
forward under every engine / fold option, against the float64 oracle, with the fp32 oracle's own error beside it; then stage by stage."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mocha_sigasia2023_amd import Generator, weights
from oracle import mocha_oracle as O
dev = torch.device("cuda:0")
sd = weights.synthetic_state_dict(4242, 2.0)
r = np.random.Generator(np.random.PCG64(103))
def spiky(n):
    X = r.standard_normal((n, 60, 24, 15)).astype(np.float32)
    for b in range(n):
        k = r.integers(0, 60 * 24 * 15, 12)
        X[b].reshape(-1)[k] = (r.uniform(30, 50, 12) * r.choice([-1.0, 1.0], 12)).astype(np.float32)
    return X
clean = r.standard_normal((32, 60, 24, 15)).astype(np.float32)
_ = spiky(32); _ = spiky(32)
src, cha = spiky(32), spiky(32)
s32 = O.to_torch_state(sd); s64 = {k: v.double() for k, v in s32.items()}
with torch.no_grad():
    st32, st64 = {}, {}
    def stages(st, S, C, store):
        ts = O.mot_embedding(st, S) + st['pos_emb'][:, :90]; tc = O.mot_embedding(st, C) + st['pos_emb'][:, :90]
        es = O.encoder(st, ts); ec = O.encoder(st, tc)
        d = O.decoder(st, es, ec); y = O.to_mot(st, d)
        store.update(tok_s=ts, enc_s=es, enc_c=ec, dec=d, Y=y)
    stages(s32, torch.from_numpy(src), torch.from_numpy(cha), st32)
    stages(s64, torch.from_numpy(src).double(), torch.from_numpy(cha).double(), st64)
print("oracle32 vs f64 per stage: " + "  ".join(f"{k} {float((st32[k].double() - st64[k]).abs().max()):.2e} (max {float(st64[k].abs().max()):.3g})" for k in st64))
for opts in ({}, {"gemm_bf16x3": 0}, {"attention_bf16x3": 0}, {"gemm_bf16x3": 0, "attention_bf16x3": 0}, {"fold_decoder": 0}, {"fold_joint": 0},
             {"gemm_bf16x3": 0, "attention_bf16x3": 0, "fold_decoder": 0, "fold_joint": 0}):
    m = Generator(device=dev).load_state_dict(sd).eval()
    for k, v in opts.items(): m.set_option(k, v)
    S, C = torch.from_numpy(src).to(dev), torch.from_numpy(cha).to(dev)
    ts = m.mot_embedding(S) + m.pos_emb[:, :90]; es = m.encoder(ts)
    tc = m.mot_embedding(C) + m.pos_emb[:, :90]; ec = m.encoder(tc)
    d = m.decoder(es, ec); y = m.to_mot(d)
    got = dict(tok_s=ts, enc_s=es, enc_c=ec, dec=d, Y=y)
    # each stage also fed with the float64 oracle's input (rounded to fp32), so that a stage's own error shows
    d_own = m.decoder(st64["enc_s"].float().to(dev), st64["enc_c"].float().to(dev)); y_own = m.to_mot(st64["dec"].float().to(dev))
    e_own = m.encoder(st64["tok_s"].float().to(dev))
    print(f"{str(opts):90s} " + "  ".join(f"{k} {float((got[k].cpu().double() - st64[k]).abs().max()):.2e}" for k in st64) +
          f"   own: enc {float((e_own.cpu().double() - st64['enc_s']).abs().max()):.2e} dec {float((d_own.cpu().double() - st64['dec']).abs().max()):.2e} to_mot {float((y_own.cpu().double() - st64['Y']).abs().max()):.2e}")
with torch.no_grad():
    d_o = O.decoder(s32, st64["enc_s"].float(), st64["enc_c"].float()); y_o = O.to_mot(s32, st64["dec"].float()); e_o = O.encoder(s32, st64["tok_s"].float())
print(f"fp32 oracle, own-stage errors: enc {float((e_o.double() - st64['enc_s']).abs().max()):.2e} dec {float((d_o.double() - st64['dec']).abs().max()):.2e} to_mot {float((y_o.double() - st64['Y']).abs().max()):.2e}")
