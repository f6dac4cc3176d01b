#!/usr/bin/env python3
"""Per-call-site kernel timing of one demo-pair step (HIP events inside libmocha_hip.so)."""
import argparse, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mocha_sigasia2023_amd import ContextBank, Generator, synthetic, synthetic_state_dict
ap = argparse.ArgumentParser(); ap.add_argument("--windows", type=int, default=585); ap.add_argument("--chunk", type=int, default=0)
ap.add_argument("--reps", type=int, default=3)
a = ap.parse_args()
dev = torch.device("cuda:0")
model = Generator(device=dev).load_state_dict(synthetic_state_dict(1777, 1.0)).eval()
if a.chunk: model.reserve(a.chunk)
for kv in filter(None, os.environ.get("MOCHA_OPTS", "").split(",")):           # e.g. MOCHA_OPTS=gemm_f16x2=1
    k, v = kv.split("="); model.set_option(k, int(v))
W = a.windows
src = torch.from_numpy(synthetic.pose_windows(1, W)).to(dev); cha = torch.from_numpy(synthetic.pose_windows(2, W)).to(dev)
m_, s_ = synthetic.cnt_norm(7); mean, std = torch.from_numpy(m_).to(dev), torch.from_numpy(s_).to(dev)
def step():
    if os.environ.get("THREE_CALLS"):
        enc_c, cnt_c, nm_c = model.encode(cha, mean, std)
        return ContextBank(model, nm_c, enc_c).characterize(src, mean, std)
    return model.characterize_pair(src, cha, mean, std)
for _ in range(2): step()
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for _ in range(5): step()
torch.cuda.synchronize(); print(f"chunk={a.chunk or 'default'} windows={W}: {(time.perf_counter()-t0)/5*1e3:.3f} ms/step")
model.profile_start()
for _ in range(a.reps): step()
p = model.profile_stop()
tot = sum(v["ms"] for v in p["kernels"].values()) / a.reps
print(f"sum of kernel time {tot:.3f} ms/step")
print(f"{'site|kernel':58s} {'n':>4s} {'ms/step':>8s} {'us/launch':>9s} {'TFLOP/s':>8s} {'GB/s':>7s}")
for k, v in sorted(p["sites"].items(), key=lambda kv: -kv[1]["ms"]):
    ms = v["ms"] / a.reps
    print(f"{k:58s} {v['launches']//a.reps:4d} {ms:8.3f} {v['ms']/v['launches']*1e3:9.1f} {v['flops']/v['ms']/1e9 if v['ms'] else 0:8.1f} {v['bytes']/v['ms']/1e6 if v['ms'] else 0:7.0f}")
