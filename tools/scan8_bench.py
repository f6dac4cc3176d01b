#!/usr/bin/env python3
"""Option "scan8" (match_scan8.hip) measured: one query against a 16 384-entry fp32 bank, planted (the query next to a row) and random
(independent N(0, 1) rows: the stage switches itself off), as the whole mocha_match call and as the streamed per-window step (configs[4])."""
import ctypes as C, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mocha_sigasia2023_amd import ContextBank, Generator, StreamingCharacterizer, synthetic, synthetic_state_dict
dev = torch.device("cuda:0")
D, NB = 90 * 256, 16384
model = Generator(layout="mixamo", device=dev).load_state_dict(synthetic_state_dict(1777, 1.0, "mixamo")).eval()
g = torch.Generator(device=dev); g.manual_seed(7)
nm = torch.randn((NB, D), device=dev, generator=g)

def state():
    st = (C.c_int32 * 2)(); model._ctx.call("mocha_scan_byte_state", 0, st, None); return st[0], st[1]

def time_query(bank, q, reps=10):
    for _ in range(3): bank.query(q)
    torch.cuda.synchronize()
    ps = []
    for _ in range(9):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): r = bank.query(q)
        e1.record(); torch.cuda.synchronize()
        ps.append(e0.elapsed_time(e1) / reps * 1e3)
    return float(np.median(ps)), r

qp = nm[4321:4322] + 0.02 * torch.randn((1, D), device=dev, generator=g)
qr = torch.randn((1, D), device=dev, generator=g)
res = {}
for on in (0, 1):
    model.set_option("scan8", on)
    for name, q in (("planted", qp), ("random", qr)):
        bank = ContextBank(model, nm, nm.view(NB, 90, 256))
        model.profile_start(); bank.query(q); prof = model.profile_stop()
        us, (dist, idx) = time_query(bank, q)
        res[(on, name)] = (int(idx[0, 0]), float(dist[0, 0]))
        print(f"scan8={on} {name:8s}: {us:7.1f} us per mocha_match call   idx {int(idx[0, 0])} dist {float(dist[0, 0]):.6f}   state {state()}   first call kernels: "
              + ", ".join(f"{k} {v['ms'] * 1e3:.1f}" for k, v in prof["kernels"].items()))
assert res[(0, "planted")] == res[(1, "planted")] and res[(0, "random")] == res[(1, "random")]
# the streamed step on a planted bank (configs[4] shape): 285 windows, bank rows planted next to the windows' own features
mean, std = synthetic.cnt_norm(7)
src = torch.from_numpy(synthetic.pose_windows(5, 285, 22)).to(dev)
nm0 = model.encode(src, mean, std)[2].reshape(285, D)
rows = torch.randperm(NB, device=dev, generator=g)[:285]
planted = nm.clone(); planted[rows] = nm0 + 0.01 * torch.randn((285, D), device=dev, generator=g)
for label, bankrows in (("planted bank", planted), ("random bank", nm)):
    for on in (0, 1):
        model.set_option("scan8", on)
        bank = ContextBank(model, bankrows, bankrows.view(NB, 90, 256))
        sc = StreamingCharacterizer(bank, mean, std, use_graph=True)
        for i in range(5): sc.step(src[i])
        torch.cuda.synchronize()
        lat = []
        for i in range(285):
            t0 = time.perf_counter(); sc.step(src[i]); torch.cuda.synchronize(); lat.append(time.perf_counter() - t0)
        lat = np.sort(np.asarray(lat)) * 1e3
        print(f"streamed step, {label}, scan8={on}: p50 {lat[142]:.3f} ms  p99 {lat[282]:.3f} ms   state {state()}")
model.set_option("scan8", 0)
