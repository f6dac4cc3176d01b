#!/usr/bin/env python3
"""One query against a 16 384-row fp32 bank whose neighbours crowd within the bf16 copy's rounding of the best row (285 planted rows at
0.2 % of the feature norm from each other): the scan through the bf16 copy + exact re-rank vs the plain fp32 scan."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mocha_sigasia2023_amd import ContextBank, Generator, synthetic, synthetic_state_dict
dev = torch.device("cuda:0")
model = Generator(device=dev).load_state_dict(synthetic_state_dict(1777, 1.0)).eval()
g = torch.Generator(device=dev); g.manual_seed(7)
nm = torch.randn((16384, 90 * 256), device=dev, generator=g)
m_, s_ = synthetic.cnt_norm(7)
src = torch.from_numpy(synthetic.pose_windows(5, 285)).to(dev)
with torch.no_grad():
    mean, std = torch.from_numpy(m_).to(dev), torch.from_numpy(s_).to(dev)
    nm0 = model.encode(src, mean, std)[2].reshape(285, -1)
    gap = (torch.cdist(nm0, nm0) + 1e30 * torch.eye(285, device=dev)).min().item()
    rows = torch.randperm(16384, device=dev, generator=g)[:285]
    planted = nm0 + (0.1 * gap / nm0.shape[1] ** 0.5) * torch.randn(nm0.shape, device=dev, generator=g)
def timed(fn, n=100):
    for _ in range(10): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
for crowded in (False, True):
    b = nm.clone()
    if crowded: b[rows] = planted
    for scan16 in (1, 0):
        model.set_option("scan16", scan16)
        bank = ContextBank(model, b, b.view(-1, 90, 256), bf16=False)
        res = []
        for qi in (0, 100, 284):
            q = nm0[qi:qi + 1].contiguous()
            _, idx = bank.query(q)
            res.append(f"{timed(lambda: bank.query(q)):.0f} us (row {int(idx[0, 0])})")
        print(f"{'crowded' if crowded else 'noise  '} bank, scan16={scan16}: " + ", ".join(res))
model.set_option("scan16", 1)
