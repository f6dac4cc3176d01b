"""Round 5 decoder changes (run with -m gpu): what the decoder derives from a bank ENTRY alone - IN(cha) and the AdaIN style MLP's
gamma / beta (net/transformer.py:49-56, 98-107) - is computed once at mocha_bank_set and read in place through frame_index; the style
MLP runs in float64; AdaIN and the attention's mapping norm are evaluated from one set of statistics (pointwise.hip).

  * cached constants == recomputed per call, bit for bit (same kernels, only the gather differs), batched / one window / streamed graph;
  * the closed-form pair against the literal two-pass order: equal to fp32 rounding on well-conditioned inputs, and both against the
    float64 oracle;
  * the float64 style MLP against the oracle's float64 MLP: gamma / beta to 1e-6 relative where fp32 GEMMs give 1e-5;
  * mocha_decoder / Generator.forward (no bank) take the same arithmetic.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from mocha_sigasia2023_amd import ContextBank, Generator, StreamingCharacterizer, synthetic, weights
from oracle import mocha_oracle as O

pytestmark = pytest.mark.gpu


def dev():
    assert torch.cuda.is_available(), "GPU tests need a ROCm device"
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def sd():
    return weights.synthetic_state_dict(515, 1.5)


@pytest.fixture(scope="module")
def model(sd):
    return Generator(device=dev()).load_state_dict(sd).eval()


def _bank(model, n=64, seed=3):
    mean, std = synthetic.cnt_norm(7)
    cha = torch.from_numpy(synthetic.pose_windows(seed, n)).to(dev())
    enc, _, nm = model.encode(cha, mean, std)
    return enc, nm, mean, std


@pytest.mark.parametrize("B", [1, 5, 40, 200])
def test_cached_bank_constants_equal_the_per_call_flow(model, B):
    """bank_dec_cache = 1 (default) against 0: the decoder reads IN(entry), gamma / beta and the entry itself through the indices instead
    of normalising / running the MLP on a gathered copy - the same kernels on the same values: Y bit for bit, same indices."""
    enc, nm, mean, std = _bank(model)
    X = torch.from_numpy(synthetic.pose_windows(100 + B, B)).to(dev())
    out = {}
    for cache in (1, 0):
        model.set_option("bank_dec_cache", cache)
        bank = ContextBank(model, nm, enc)                                     # the option takes effect at mocha_bank_set
        model.profile_start()
        Y, idx = bank.characterize(X, mean, std, return_index=True)
        sites = model.profile_stop()["sites"]
        out[cache] = (Y.clone(), idx.clone(), sites)
    model.set_option("bank_dec_cache", 1)
    assert torch.equal(out[1][1], out[0][1])
    if B > 16:
        assert torch.equal(out[1][0], out[0][0]), float((out[1][0] - out[0][0]).abs().max())
    else:
        # up to 16 rows the per-call style MLP runs on the few-rows float64 kernel (another summation order than the bank build's tiled one):
        # gamma / beta agree to float64 rounding, i.e. to the last fp32 bit except on a rounding boundary
        assert float((out[1][0] - out[0][0]).abs().max()) <= 1e-6 * max(1.0, float(out[0][0].abs().max()))
    names = {c: {k.split("|")[0] for k in out[c][2]} for c in (0, 1)}           # profile keys are "site|kernel"
    assert "dec.in_cha" in names[0] and "dec.style1" in names[0]               # recomputed per call ...
    assert "dec.in_cha" not in names[1] and "dec.style1" not in names[1]       # ... and not at all with the cache
    assert len(set(idx.cpu().tolist())) > 1 or B == 1


def test_cached_constants_in_the_streamed_step_and_after_a_new_bank(model):
    """The captured per-window step (mocha_step_graph) bakes the bank's constant tables in: a new bank moves the generation and the
    step re-captures; results equal the batched call's windows."""
    enc, nm, mean, std = _bank(model, 48, seed=5)
    X = torch.from_numpy(synthetic.pose_windows(77, 6)).to(dev())
    bank = ContextBank(model, nm, enc)
    Yb, ib = bank.characterize(X, mean, std, return_index=True)
    sc = StreamingCharacterizer(bank, mean, std, use_graph=True)
    for w in range(3):
        y, i = sc.step(X[w])
        assert int(i.item()) == int(ib[w]) and float((y - Yb[w]).abs().max()) < 1e-5 * max(1.0, float(Yb.abs().max()))
    enc2, nm2, _, _ = _bank(model, 31, seed=6)                                 # another bank, other size: tables replaced
    bank2 = ContextBank(model, nm2, enc2)
    Yb2, ib2 = bank2.characterize(X, mean, std, return_index=True)
    sc2 = StreamingCharacterizer(bank2, mean, std, use_graph=True)
    for w in range(3, 6):
        y, i = sc2.step(X[w])
        assert int(i.item()) == int(ib2[w]) and float((y - Yb2[w]).abs().max()) < 1e-5 * max(1.0, float(Yb2.abs().max()))


def test_closed_form_adain_against_the_literal_order_and_float64(model, sd):
    """adain_closed_form = 1 against 0 on white-noise windows (well conditioned at this gain): the two evaluations of IN(AdaIN(x)) agree
    to fp32 rounding, and both are within 1e-4 of the float64 oracle; the closed form is the nearer one or equal within noise."""
    S = synthetic.pose_windows(31, 24); C = synthetic.pose_windows(32, 24)
    s32 = O.to_torch_state(sd); s64 = {k: v.double() for k, v in s32.items()}
    with torch.no_grad():
        Y64 = O.generator_forward(s64, torch.from_numpy(S).double(), torch.from_numpy(C).double())
    res = {}
    for closed in (1, 0):
        model.set_option("adain_closed_form", closed)
        res[closed] = model(torch.from_numpy(S).to(dev()), torch.from_numpy(C).to(dev())).cpu().double()
    model.set_option("adain_closed_form", 1)
    scale = max(1.0, float(Y64.abs().max()))
    d = float((res[1] - res[0]).abs().max()); e1 = float((res[1] - Y64).abs().max()); e0 = float((res[0] - Y64).abs().max())
    print(f"[decoder] closed form vs literal order: {d:.2e}; against float64: closed {e1:.2e}, literal {e0:.2e} (max |Y| {scale:.3g})")
    assert d < 2e-5 * scale and e1 < 1e-4 * scale and e0 < 1e-4 * scale


def test_float64_style_mlp(model, sd):
    """gamma / beta of the AdaIN style MLP (net/transformer.py:100-107) against the oracle's float64 MLP on the same fp32 entries: the
    float64 kernel (mocha_linear_f64, float64 token mean) is rounded once (<= 1 ulp of fp32), the fp32 engines are ~10x further."""
    enc, nm, mean, std = _bank(model, 40, seed=9)
    s64 = {k: v.double() for k, v in O.to_torch_state(sd).items()}
    e64 = enc.cpu().double()
    ref = []
    for l in range(2):
        p = f"decoder.layers.{l}.0"
        ref.append(F.linear(F.leaky_relu(F.linear(e64.mean(1), s64[f"{p}.style.2.weight"], s64[f"{p}.style.2.bias"]), 0.2),
                            s64[f"{p}.style.4.weight"], s64[f"{p}.style.4.bias"]))
    ref = torch.cat(ref, 1)                                                    # (40, 1024): [gamma0 | beta0 | gamma1 | beta1]
    errs = {}
    for f64 in (1, 0):
        model.set_option("style_f64", f64)
        got = model.style_constants(enc).cpu().double()
        errs[f64] = float(((got - ref).abs() / (ref.abs() + 1.0)).max())
    model.set_option("style_f64", 1)
    print(f"[decoder] style MLP against float64: float64 kernel {errs[1]:.2e}, fp32 engines {errs[0]:.2e}")
    assert errs[1] < 1.5e-7 and errs[0] < 1e-4 and errs[1] < errs[0]
