#!/usr/bin/env python3
"""A/B of the decoder attention from pre-split key / value images (option "attention_kv": 0 = mocha_attention_x3<256> on the fp32 rows)
at the demo step's batch (585 windows) and around it: whole decoder call, and per kernel (HIP events) the attention and the
instance norm that writes the images."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mocha_sigasia2023_amd import Generator, synthetic, synthetic_state_dict
dev = torch.device("cuda:0")
model = Generator(device=dev).load_state_dict(synthetic_state_dict(1777, 1.0)).eval()
def timed(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6
for B in (64, 128, 256, 585, 1024):
    tok = torch.from_numpy(synthetic.token_features(5, B)).to(dev); cha = torch.from_numpy(synthetic.token_features(6, B)).to(dev)
    for kv in (0, 1, 2, 0, 1, 2):                    # 2 = the six-wave-per-head-pair instance
        model.set_option("attention_kv", 1 if kv else 0)
        model.set_option("attention_kv_pairs", 1 if kv == 2 else 0)
        t = timed(lambda: model.decoder(tok, cha))
        model.profile_start()
        for _ in range(10): model.decoder(tok, cha)
        k = model.profile_stop()["sites"]
        pick = {s.split("|")[0] + ":" + s.split("|")[1][:28]: v["ms"] / v["launches"] * 1e3 for s, v in k.items() if s.startswith("dec.attn") or s.startswith("dec.in_cha")}
        print(f"B={B:5d} attention_kv={kv}: decoder {t:8.1f} us   " + "  ".join(f"{a} {b:7.1f} us" for a, b in sorted(pick.items())), flush=True)
