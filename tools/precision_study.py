#!/usr/bin/env python3
"""CPU-only study (no GPU, runs in the build container): WHY the structured-input cases of tests/test_structured_inputs.py cannot be
held to 1e-4 against the fp32 oracle, and WHICH operations of the decoder produce the error.

Part 1 - the reference arithmetic against ITSELF.  The fp32 oracle (= the reference's torch ops, op for op) evaluated under settings
a user of the reference may legitimately choose - batch 1 / 8 / all windows per call (test_fullframework.py:465 decodes one window at
a time, collect_CVAE_feature_action.py:167 encodes 32), 1 / 8 intra-op threads - compared with each other and with the float64
evaluation.  If two fp32 runs of the reference differ by more than 1e-4, no third implementation can be within 1e-4 of "the" fp32
reference; the meaningful bound is then the float64 result.  When /root/reference is importable the same is done with the reference's
own Generator module (build container only).

Part 2 - one operation at a time in fp32, everything else in float64 (decoder + to_mot, fed the float64 encoder features): the
contribution of each operation's rounding to |Y - Y64|.  This is what tells which launches of the HIP path would have to carry more
than fp32 to get nearer the float64 result (net/transformer.py:13-20, 49-56, 63-76, 98-121).

Usage: python tools/precision_study.py [gain ...]   (default gains 2.0)
"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mocha_sigasia2023_amd import synthetic, weights          # noqa: E402
from oracle import mocha_oracle as O                            # noqa: E402  (checker-side tool)

torch.set_grad_enabled(False)


def cases(seed=0):
    r = np.random.Generator(np.random.PCG64(103 + seed))

    def spiky(n):
        X = r.standard_normal((n, 60, 24, 15)).astype(np.float32)
        for b in range(n):
            k = r.integers(0, 60 * 24 * 15, 12)
            X[b].reshape(-1)[k] = (r.uniform(30, 50, 12) * r.choice([-1.0, 1.0], 12)).astype(np.float32)
        return X
    out = {}
    out["outliers in both"] = (spiky(32), spiky(32))
    X = synthetic.pose_windows(104 + seed, 48)
    out["cha == src"] = (X, X)
    r2 = np.random.Generator(np.random.PCG64(101 + seed))
    out["static pose"] = (np.repeat(r2.standard_normal((24, 1, 24, 15)).astype(np.float32), 60, axis=1),
                          np.repeat(r2.standard_normal((24, 1, 24, 15)).astype(np.float32), 60, axis=1))
    out["white noise"] = (synthetic.pose_windows(1 + seed, 32), synthetic.pose_windows(2 + seed, 32))
    return out


def fwd(st, S, C, batch):
    return torch.cat([O.generator_forward(st, S[s:s + batch], C[s:s + batch]) for s in range(0, len(S), batch)])


def part1(sd, gain):
    s32 = O.to_torch_state(sd); s64 = {k: v.double() for k, v in s32.items()}
    ref = None
    if os.path.isdir("/root/reference"):
        try:
            ref = _reference_model(sd)
        except Exception as e:                                       # noqa: BLE001
            print(f"(reference not importable here: {e})")
    print(f"== part 1, gain {gain}: fp32 evaluations of the reference arithmetic against each other (max abs over Y)")
    for name, (S, C) in cases().items():
        S32, C32 = torch.from_numpy(S), torch.from_numpy(C)
        Y64 = fwd(s64, S32.double(), C32.double(), len(S))
        runs = {}
        for thr in (8, 1):
            torch.set_num_threads(thr)
            for b in (len(S), 8, 1):
                runs[f"b{b}/t{thr}"] = fwd(s32, S32, C32, b)
        torch.set_num_threads(8)
        base = runs[f"b{len(S)}/t8"]
        line = f"{name:18s} max|Y| {float(Y64.abs().max()):5.2f}  |o32-f64| " + " ".join(
            f"{k}:{float((v.double() - Y64).abs().max()):.2e}" for k, v in runs.items())
        line += "   |o32 - o32(b_all,t8)| " + " ".join(f"{k}:{float((v - base).abs().max()):.2e}" for k, v in runs.items() if v is not base)
        print(line)
        if ref is not None:
            ya = ref(S32, C32); y1 = torch.cat([ref(S32[i:i + 1], C32[i:i + 1]) for i in range(len(S))])
            print(f"{'':18s} reference module: |ref(b_all) - oracle(b_all)| {float((ya - base).abs().max()):.2e}   |ref(b1) - ref(b_all)| "
                  f"{float((y1 - ya).abs().max()):.2e}   |ref(b1) - f64| {float((y1.double() - Y64).abs().max()):.2e}")


def _reference_model(sd):
    """The reference's own Generator with the synthetic weights (build container only; nothing of it is stored)."""
    cwd = os.getcwd(); os.chdir("/root/reference")
    try:
        for p in (".", "./net", "./etc"):
            if p not in sys.path:
                sys.path.append(p)
        from utils import get_config                                  # type: ignore
        from model import Generator                                   # type: ignore
        m = Generator(get_config("./configs/config.yaml")["model"]).eval()
    finally:
        os.chdir(cwd)
    missing = m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=False)
    assert not missing.missing_keys, missing
    return m


# ------------------------------------------------------------------------------------------------ part 2: one op in fp32
class P:
    """Precision policy: every op runs in float64 except those named in `low` (inputs rounded to fp32, computed in fp32)."""

    def __init__(self, low=()):
        self.low = set(low)

    def run(self, name, fn, *args):
        if name in self.low or "*" in self.low:
            return fn(*(a.float() if torch.is_tensor(a) else a for a in args)).double()
        return fn(*(a.double() if torch.is_tensor(a) else a for a in args))


def mvn(x):                                                            # (B, N, C) over tokens, net/transformer.py:13-20
    m = x.mean(1, keepdim=True); s = x.std(1, keepdim=True)
    return (x - m) / (s + 1e-5)


def decoder_mixed(st, x, sty, pol, heads=4):
    W = lambda k: st[k]                                                # noqa: E731
    for l in range(2):
        p = f"decoder.layers.{l}"
        s = pol.run("style", lambda a, w1, b1, w2, b2: F.linear(F.leaky_relu(F.linear(a.mean(1), w1, b1), 0.2), w2, b2),
                    sty, W(f"{p}.0.style.2.weight"), W(f"{p}.0.style.2.bias"), W(f"{p}.0.style.4.weight"), W(f"{p}.0.style.4.bias"))
        g, b = s.chunk(2, 1)
        xin = pol.run("adain.norm", mvn, x)
        x = pol.run("adain.affine", lambda a, g_, b_: (1 + g_[:, None]) * a + b_[:, None], xin, g, b)
        qi = pol.run("attn.in_q", mvn, x)
        ki = pol.run("attn.in_k", mvn, sty)
        q = pol.run("to_q", F.linear, qi, W(f"{p}.1.to_q.1.weight"))
        k = pol.run("to_k", F.linear, ki, W(f"{p}.1.to_k.1.weight"))
        v = pol.run("to_v", F.linear, sty, W(f"{p}.1.to_v.weight"))
        B, N, inner = q.shape; dh = inner // heads
        q, k, v = (t.view(B, -1, heads, dh).permute(0, 2, 1, 3) for t in (q, k, v))
        dots = pol.run("dots", lambda a, b_: torch.matmul(a, b_.transpose(-1, -2)) * dh ** -0.5, q, k)
        pr = pol.run("softmax", lambda a: a.softmax(-1), dots)
        o = pol.run("attn.v", torch.matmul, pr, v).permute(0, 2, 1, 3).reshape(B, N, inner)
        x = pol.run("to_out", lambda a, w, b_, r: F.linear(a, w, b_) + r, o, W(f"{p}.1.to_out.0.weight"), W(f"{p}.1.to_out.0.bias"), x)
        h = pol.run("ff1", lambda a, w, b_: F.gelu(F.linear(a, w, b_)), x, W(f"{p}.2.net.0.weight"), W(f"{p}.2.net.0.bias"))
        x = pol.run("ff2", lambda a, w, b_, r: F.linear(a, w, b_) + r, h, W(f"{p}.2.net.3.weight"), W(f"{p}.2.net.3.bias"), x)
    return x


OPS = ["style", "adain.norm", "adain.affine", "attn.in_q", "attn.in_k", "to_q", "to_k", "to_v", "dots", "softmax", "attn.v", "to_out", "ff1", "ff2"]


def part2(sd, gain):
    s32 = O.to_torch_state(sd); s64 = {k: v.double() for k, v in s32.items()}
    both = {k: v for k, v in s64.items()}
    print(f"== part 2, gain {gain}: decoder with ONE operation in fp32 (rest float64, float64 encoder features in): max |Y - Y64|")
    for name, (S, C) in cases().items():
        es, _ = O.encode(s64, torch.from_numpy(S).double()); ec, _ = O.encode(s64, torch.from_numpy(C).double())
        d64 = decoder_mixed(both, es, ec, P())
        assert float((d64 - O.decoder(s64, es, ec)).abs().max()) < 1e-9
        Y64 = O.to_mot(s64, d64)
        row = {}
        for op in OPS + ["*"]:
            d = decoder_mixed(both, es, ec, P([op]))
            row[op] = (float((d - d64).abs().max()), float((O.to_mot(s64, d) - Y64).abs().max()))
        # all fp32 except one op kept in float64: how much of the total that op is responsible for
        keep = {}
        for op in OPS:
            d = decoder_mixed(both, es, ec, P([o for o in OPS if o != op]))
            keep[op] = float((O.to_mot(s64, d) - Y64).abs().max())
        print(f"-- {name} (max|dec| {float(d64.abs().max()):.3g}, max|Y| {float(Y64.abs().max()):.3g})")
        print("   only this op fp32 -> |Y-Y64|: " + "  ".join(f"{op} {row[op][1]:.1e}" for op in OPS) + f"   ALL fp32 {row['*'][1]:.1e}")
        print("   all fp32 but this op     : " + "  ".join(f"{op} {keep[op]:.1e}" for op in OPS))


if __name__ == "__main__" and os.environ.get("PRECISION_PART12", "1") == "1":
    gains = [float(a) for a in sys.argv[1:]] or [2.0]
    for g in gains:
        sd = weights.synthetic_state_dict(4242, g)
        part1(sd, g)
        part2(sd, g)


# ------------------------------------------------------------------------------------------------ part 3: the fused-norm form
def decoder_fused_norms(st32, st64, x, sty, heads=4, style64=True, fused=True, stats64=False):
    """fp32 decoder in which (a) the style MLP runs in float64 from a float64 token mean and (b) AdaIN and the instance norm of the
    attention's query input are evaluated from ONE set of statistics of x:  with m, s = mean / unbiased std of x over the tokens,
    AdaIN(x) = (1+g)(x-m)/(s+eps) + b, whose own token mean is exactly b and whose std is |1+g| s/(s+eps), hence
    IN(AdaIN(x)) = (1+g)(x-m) / (|1+g| s + eps (s+eps)) - no cancellation against b.  Everything else fp32 as the reference."""
    eps = 1e-5
    for l in range(2):
        p = f"decoder.layers.{l}"
        if style64:
            s_ = F.linear(F.leaky_relu(F.linear(sty.double().mean(1), st64[f"{p}.0.style.2.weight"], st64[f"{p}.0.style.2.bias"]), 0.2),
                          st64[f"{p}.0.style.4.weight"], st64[f"{p}.0.style.4.bias"])
        else:
            s_ = F.linear(F.leaky_relu(F.linear(sty.mean(1), st32[f"{p}.0.style.2.weight"], st32[f"{p}.0.style.2.bias"]), 0.2),
                          st32[f"{p}.0.style.4.weight"], st32[f"{p}.0.style.4.bias"]).double()
        g, b = s_.chunk(2, 1)                                           # float64 (B, 256)
        m = x.mean(1, keepdim=True); s = x.std(1, keepdim=True)        # fp32 statistics (two-pass)
        if fused and stats64:
            # statistics, centring and coefficients in float64 from the fp32 activations; results rounded to fp32 once
            xd = x.double(); md = xd.mean(1, keepdim=True); sd = xd.std(1, keepdim=True)
            g1 = (1.0 + g)[:, None]
            xcd = xd - md
            xa = (g1 / (sd + eps) * xcd + b[:, None]).float()
            qi = (g1 / (g1.abs() * sd + eps * (sd + eps)) * xcd).float()
        elif fused:
            g1 = (1.0 + g)[:, None]                                     # float64 per (window, channel) coefficients, rounded once
            a1 = (g1 / (s.double() + eps)).float(); a2 = (g1 / (g1.abs() * s.double() + eps * (s.double() + eps))).float()
            xc = x - m
            xa = a1 * xc + b[:, None].float()
            qi = a2 * xc
        else:
            xa = (1 + g.float())[:, None] * ((x - m) / (s + eps)) + b.float()[:, None]
            qi = mvn(xa)
        ki = mvn(sty.double()).float() if stats64 else mvn(sty)
        q = F.linear(qi, st32[f"{p}.1.to_q.1.weight"]); k = F.linear(ki, st32[f"{p}.1.to_k.1.weight"]); v = F.linear(sty, st32[f"{p}.1.to_v.weight"])
        B, N, inner = q.shape; dh = inner // heads
        q, k, v = (t.view(B, -1, heads, dh).permute(0, 2, 1, 3) for t in (q, k, v))
        pr = (torch.matmul(q, k.transpose(-1, -2)) * dh ** -0.5).softmax(-1)
        o = torch.matmul(pr, v).permute(0, 2, 1, 3).reshape(B, N, inner)
        x = F.linear(o, st32[f"{p}.1.to_out.0.weight"], st32[f"{p}.1.to_out.0.bias"]) + xa
        x = F.linear(F.gelu(F.linear(x, st32[f"{p}.2.net.0.weight"], st32[f"{p}.2.net.0.bias"])), st32[f"{p}.2.net.3.weight"], st32[f"{p}.2.net.3.bias"]) + x
    return x


def part3(gains=(2.0,), seeds=range(8)):
    print("== part 3: fp32 decoder variants fed the fp32 oracle's encoder features: max |Y - Y64| (Y64 = float64 end to end)")
    print(f"{'gain':>4} {'seed':>4} {'case':18s} {'max|Y|':>7} {'oracle32':>9} {'style64':>9} {'fused':>9} {'both':>9} {'+stats64':>9}")
    worst = {}
    for g in gains:
        for sw in seeds:
            sd = weights.synthetic_state_dict(4242 + sw, g)
            s32 = O.to_torch_state(sd); s64 = {k: v.double() for k, v in s32.items()}
            for name, (S, C) in cases(sw).items():
                S32, C32 = torch.from_numpy(S), torch.from_numpy(C)
                Y64 = fwd(s64, S32.double(), C32.double(), len(S))
                es, _ = O.encode(s32, S32); ec, _ = O.encode(s32, C32)
                res = [float((O.to_mot(s32, O.decoder(s32, es, ec)).double() - Y64).abs().max())]
                for style64, fused, st64f in ((True, False, False), (False, True, False), (True, True, False), (True, True, True)):
                    d = decoder_fused_norms(s32, s64, es, ec, style64=style64, fused=fused, stats64=st64f)
                    res.append(float((O.to_mot(s32, d).double() - Y64).abs().max()))
                print(f"{g:4.1f} {sw:4d} {name:18s} {float(Y64.abs().max()):7.2f} " + " ".join(f"{e:9.2e}" for e in res), flush=True)
                w = worst.setdefault(name, [0, 0, 0, 0, 0])
                for i, e in enumerate(res):
                    w[i] = max(w[i], e)
    for name, w in worst.items():
        print(f"worst      {name:18s}         " + " ".join(f"{e:9.2e}" for e in w))


if __name__ == "__main__" and os.environ.get("PRECISION_PART3", "1") == "1":
    part3()
