"""The identity behind mocha_adain's closed form (pointwise.hip), checked in float64 on the CPU against the oracle's
literal evaluation (net/transformer.py:108-113 followed by :49-56): with m, s the token mean / unbiased std of a channel of x,

    AdaIN(x) = (1 + g) (x - m) / (s + eps) + b           has token mean exactly b and std |1 + g| s / (s + eps), hence
    IN(AdaIN(x)) = (1 + g) (x - m) / (|1 + g| s + eps (s + eps)).

Equal to rounding (1e-12) wherever the literal order is well-conditioned - including negative and tiny gains, large offsets, and
zero-variance channels (0) - and the case that motivates it: in fp32 the literal order loses the channel to the cancellation
against b, the closed form does not."""
import numpy as np
import torch

from oracle import mocha_oracle as O

EPS = 1e-5


def closed_form(x, g, b):
    m = x.mean(1, keepdim=True); s = x.std(1, keepdim=True)
    g1 = (1.0 + g)[:, None]
    xad = g1 * (x - m) / (s + EPS) + b[:, None]
    qin = g1 * (x - m) / (g1.abs() * s + EPS * (s + EPS))
    return xad, qin


def literal(x, g, b):
    n = O.mean_variance_norm(x.permute(0, 2, 1)).permute(0, 2, 1)
    xad = (1.0 + g)[:, None] * n + b[:, None]
    return xad, O.mean_variance_norm(xad.permute(0, 2, 1)).permute(0, 2, 1)


def test_closed_form_equals_the_literal_order_in_float64():
    r = np.random.Generator(np.random.PCG64(5))
    x = torch.from_numpy(r.standard_normal((6, 90, 256))) * torch.from_numpy(r.uniform(0.01, 30.0, (6, 1, 256)))
    g = torch.from_numpy(r.standard_normal((6, 256)) * 2.0)
    b = torch.from_numpy(r.standard_normal((6, 256)) * 10.0)
    g[:, 0:8] = -1.0 + torch.from_numpy(r.uniform(-1e-3, 1e-3, (6, 8)))        # 1 + g tiny, either sign
    g[:, 8:12] = -3.0                                                           # negative gain: the sign survives the norm
    x[:, :, 12:16] = 4.25                                                       # zero variance
    xa_c, q_c = closed_form(x, g, b)
    xa_l, q_l = literal(x, g, b)
    assert float((xa_c - xa_l).abs().max()) < 1e-12
    ok = torch.ones(256, dtype=torch.bool); ok[0:8] = False                    # tiny 1 + g: compared separately below
    assert float((q_c - q_l)[:, :, ok].abs().max()) < 1e-9
    # zero variance: the closed form gives exactly 0; the literal order the rounding of mean(b + 0) divided by eps
    assert float(q_c[:, :, 12:16].abs().max()) == 0.0 and float(q_l[:, :, 12:16].abs().max()) < 1e-8
    # tiny 1 + g: even float64 loses digits to b in the literal order (1e-3 against |b| ~ 10: ~1e-12 relative); still equal to 1e-8
    assert float((q_c - q_l)[:, :, 0:8].abs().max()) < 1e-8


def test_the_literal_order_loses_small_gain_channels_in_fp32_and_the_closed_form_does_not():
    r = np.random.Generator(np.random.PCG64(6))
    x = torch.from_numpy(r.standard_normal((4, 90, 256)))
    g = torch.from_numpy(r.standard_normal((4, 256)))
    b = torch.from_numpy(r.standard_normal((4, 256)) * 10.0)
    g[:, :32] = -1.0 + 2e-4                                                    # |1 + g| = 2e-4 against |b| ~ 10
    _, q64 = closed_form(x, g, b)
    _, q_lit32 = literal(x.float(), g.float(), b.float())
    _, q_cf32 = closed_form(x.float(), g.float(), b.float())
    e_lit = float((q_lit32.double() - q64)[:, :, :32].abs().max())
    e_cf = float((q_cf32.double() - q64)[:, :, :32].abs().max())
    assert e_cf < 2e-3 < e_lit, (e_cf, e_lit)          # fp32 rounding of 1 + g itself (1e-7 / 2e-4) bounds the closed form; the literal order is ~100x worse
    assert float((q_cf32.double() - q64)[:, :, 32:].abs().max()) < 1e-5
