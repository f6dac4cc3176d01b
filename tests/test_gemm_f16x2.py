"""The two-plane fp16 / three-pass GEMM engine (csrc/gemm_h2.hip, option "gemm_f16x2", default off; run with -m gpu).

x = fp16(S x) + fp16(residual) carries 22 bits per operand, a b ~ a0 b0 + a0 b1 + a1 b0 on v_mfma_f32_32x32x16_f16 with fp32 accumulation.
Per product that is 2^-22, not the 2^-24 of an fp32 multiply - but against float64 a K-long dot product comes out MORE accurate than on
either fp32 engine of the library, because the accumulator's own roundings dominate and three passes have half as many as six
(tools/f16x2_sim.py).  Checked here:

  * GEMM error against float64 <= the exact-f32 MFMA kernel's (every shape) and <= the three-plane bf16 engine's (K >= 256: the step's
    shapes), with bias, on 64-wide tiles and short K loops; operands with large / tiny magnitudes (the per-launch activation scale and the per-row weight scales
    are exact powers of two) and 40-sigma outlier channels;
  * the scale does not change the result while nothing leaves fp16's normal range: x and 2^k x give bit-identical y / 2^k;
  * one scale per WINDOW (the path's independent unit): windows six decades apart are each as accurate as on the fp32 engine, and a window's
    result does not depend on what else is in the batch (bit for bit); the documented limit - rows more than five decades below THEIR
    window's largest lose digits - measured, bounded;
  * the whole network with the option on: poses within the north-star 1e-4 of the fp32 oracle (golden fixtures), same matches and
    poses within 2e-5 of the default engine at the demo pair's full size, against a bank, and through Generator.forward.
"""
import numpy as np
import pytest
import torch

from mocha_sigasia2023_amd import ContextBank, Generator, synthetic, weights

pytestmark = pytest.mark.gpu


def dev():
    assert torch.cuda.is_available(), "GPU tests need a ROCm device"
    return torch.device("cuda:0")


def _errs(model, x, w, b, engines=(("f32", 1), ("x3", 2), ("h2", 3))):
    ref = x.double() @ w.double().T
    if b is not None:
        ref = ref + b.double()
    out = {}
    for name, engine in engines:
        y = model.linear(x, w, b, engine=engine)
        d = y.double() - ref
        out[name] = (float(d.abs().max()), float(d.pow(2).mean().sqrt()))
    return out, float(ref.abs().max())


@pytest.mark.parametrize("M,N,K,bias", [(105300, 256, 512, True), (105300, 512, 256, False), (105300, 256, 1024, True), (30001, 256, 1280, True),
                                        (200000, 128, 32, True), (98305, 384, 64, False), (210001, 64, 320, True), (105300, 192, 256, False),
                                        (11520, 256, 512, True), (11520, 192, 256, False), (4000, 1536, 256, False)])
def test_two_plane_fp16_engine_against_float64(M, N, K, bias):
    g = torch.Generator(device="cpu").manual_seed(M + N + K)
    x = torch.randn((M, K), generator=g, dtype=torch.float32).to(dev())
    w = (torch.randn((N, K), generator=g, dtype=torch.float32) / np.sqrt(K)).to(dev())
    b = torch.randn((N,), generator=g, dtype=torch.float32).to(dev()) if bias else None
    model = Generator(layout="mixamo", device=dev())
    engines = (("f32", 1), ("x3", 2), ("h2", 3)) if M >= 20000 or N % 128 else (("f32", 1), ("h2", 3))
    err, scale = _errs(model, x, w, b, engines)
    print(f"[f16x2] {M} x {N} x {K}: rms error  " + "  ".join(f"{k} {v[1]:.2e}" for k, v in err.items()))
    assert err["h2"][1] <= err["f32"][1] * 1.02 + 1e-9, (err, scale)
    assert err["h2"][0] <= err["f32"][0] * 1.5 + 1e-9, (err, scale)
    if "x3" in err and K >= 256:        # short K loops (two / four steps): the planes' 2^-22 shows beside so few accumulator roundings - still below the exact pipe
        assert err["h2"][1] <= err["x3"][1] * 1.05 + 1e-9, (err, scale)


@pytest.mark.parametrize("xs,ws", [(1e4, 1e-3), (3e-5, 7e2), (1.0, 1.0)])
def test_scales_are_exact(xs, ws):
    """Large / tiny operands: the activation scale (from the measured bound) and the weight scales are powers of two - the error relative to
    the output's size does not move; and a power-of-two multiple of x gives the same bits."""
    M, N, K = 40000, 256, 512
    g = torch.Generator(device="cpu").manual_seed(11)
    x = (torch.randn((M, K), generator=g) * xs).float().to(dev())
    w = (torch.randn((N, K), generator=g) * ws / np.sqrt(K)).float().to(dev())
    model = Generator(layout="mixamo", device=dev())
    err, scale = _errs(model, x, w, None, (("f32", 1), ("h2", 3)))
    assert err["h2"][1] <= err["f32"][1] * 1.02 and np.isfinite(err["h2"][0]), (err, scale)
    y1 = model.linear(x, w, None, engine=3)
    y2 = model.linear(x * 64.0, w * 0.125, None, engine=3)
    assert torch.equal(y1 * 8.0, y2)


def test_outlier_channels_and_the_range_limit_inside_a_window():
    M, N, K = 40000, 256, 512
    g = torch.Generator(device="cpu").manual_seed(5)
    x = torch.randn((M, K), generator=g); x[:, ::7] *= 40.0
    w = torch.randn((N, K), generator=g) / np.sqrt(K)
    model = Generator(layout="mixamo", device=dev())
    err, scale = _errs(model, x.float().to(dev()), w.float().to(dev()), None)
    assert err["h2"][1] <= err["f32"][1] * 1.02 and err["h2"][1] <= err["x3"][1] * 1.05, (err, scale)
    # One activation scale per WINDOW (here: blocks of 1 024 rows).  Windows six decades apart: every row as good as on the fp32 engine.
    wd = w.float().to(dev())
    blk = torch.logspace(-4, 2, (M + 1023) // 1024).repeat_interleave(1024)[:M].unsqueeze(1)
    rel = _row_errors(model, (torch.randn((M, K), generator=g) * blk).float().to(dev()), wd)
    print(f"[f16x2] windows six decades apart: worst row's relative rms error  f32 {rel['f32'].max():.2e}  h2 {rel['h2'].max():.2e}")
    assert rel["h2"].max() <= rel["f32"].max() * 1.05
    # Rows six decades apart INSIDE a window: the smallest rows keep fewer digits than on the fp32 engines - the documented limit
    # (rows down to 1e-4 of their window's largest are unaffected)
    inw = torch.logspace(-4, 2, 1024).repeat((M + 1023) // 1024)[:M].unsqueeze(1)
    rel = _row_errors(model, (torch.randn((M, K), generator=g) * inw).float().to(dev()), wd)
    small = (inw.squeeze(1) < 1e-2).numpy()
    print(f"[f16x2] rows six decades apart inside a window: worst row  f32 {rel['f32'].max():.2e}  h2 {rel['h2'].max():.2e}; "
          f"rows within four decades of the window's largest: f32 {rel['f32'][~small].max():.2e}  h2 {rel['h2'][~small].max():.2e}")
    assert rel["h2"][~small].max() <= rel["f32"][~small].max() * 1.05             # the upper four decades: as good as fp32
    assert rel["h2"].max() < 2e-5                                                  # the lowest rows: degraded, bounded


def _row_errors(model, x, w):
    ref = x.double() @ w.double().T
    out = {}
    for name, engine in (("f32", 1), ("h2", 3)):
        d = model.linear(x, w, None, engine=engine).double() - ref
        out[name] = (d.pow(2).mean(1).sqrt() / ref.pow(2).mean(1).sqrt()).cpu().numpy()
    return out


def test_activation_bound_of_an_unaligned_operand():
    """mocha_absmax reads 16 bytes at a time from the first aligned element on: an operand that starts 4 bytes off gives the same result."""
    M, N, K = 8192, 256, 256
    g = torch.Generator(device="cpu").manual_seed(3)
    flat = torch.randn((M * K + 5,), generator=g).to(dev())
    flat[1] = 300.0; flat[M * K] = -700.0                          # the largest magnitudes sit in the unaligned head and tail
    xv = flat[1:1 + M * K].view(M, K)
    w = (torch.randn((N, K), generator=g) / 16).to(dev())
    model = Generator(layout="mixamo", device=dev())
    assert xv.data_ptr() % 16 == 4
    y1 = model.linear(xv, w, None, engine=3); y2 = model.linear(xv.clone(), w, None, engine=3)
    assert torch.equal(y1, y2)
    ref = xv.double() @ w.double().T
    assert float((y1.double() - ref).abs().max()) < 1e-5 * float(ref.abs().max())


def test_rejected_shapes():
    model = Generator(layout="mixamo", device=dev())
    with pytest.raises(RuntimeError, match="outside the f16x2 engine"):
        model.linear(torch.randn((256, 64), device=dev()), torch.randn((100, 64), device=dev()), None, engine=3)      # N % 64


@pytest.mark.timeout(900)
def test_network_with_the_option_on():
    """Demo pair at full size (585 + 585 windows), against a bank, and Generator.forward: the option moves the encoder's, decoder's and
    to_mot's (and, for z-scored input, the embedding's) plane GEMMs to mocha_gemm_h2; same matches, poses within 2e-5 of the default engine."""
    W, V = 585, 22
    sd = weights.synthetic_state_dict(1777, 1.0, "mixamo")
    model = Generator(layout="mixamo", device=dev()).load_state_dict(sd).eval()
    src = torch.from_numpy(synthetic.pose_windows(1777, W, V)).to(dev())
    cha = torch.from_numpy(synthetic.pose_windows(1778, W, V)).to(dev())
    mean, std = synthetic.cnt_norm(7)
    outs = {}
    for flag in (0, 1):
        model.set_option("gemm_f16x2", flag)
        model.profile_start()
        Y, idx = model.characterize_pair(src, cha, mean, std, return_index=True)
        prof = model.profile_stop()
        names = set(prof["kernels"])
        assert ("mocha_gemm_h2" in names) == bool(flag), names
        if flag:
            sites = {k.split("|")[0] for k in prof["sites"] if k.endswith("|mocha_gemm_h2")}
            assert {"emb.joint_block", "emb.gcn_body", "emb.tcn_body", "enc.qkv", "xf.out_proj", "xf.ff1", "xf.ff2", "dec.q", "mot.gcn_body", "mot.tcn_body",
                    "mot.gcn_joint", "mot.tcn_joint"} <= sites, sites
        enc, _, nm = model.encode(cha, mean, std)
        bank = ContextBank(model, nm, enc)
        Yb, ib = bank.characterize(src, mean, std, return_index=True)
        Yf = model(src[:64], cha[:64])
        outs[flag] = (Y.cpu().numpy(), idx.cpu().numpy(), Yb.cpu().numpy(), ib.cpu().numpy(), Yf.cpu().numpy())
    model.set_option("gemm_f16x2", 0)
    for a, b in ((0, 1), (2, 3)):
        same = outs[1][b] == outs[0][b]
        assert same.mean() > 0.99
        assert np.abs(outs[1][a][same] - outs[0][a][same]).max() < 2e-5
    assert np.abs(outs[1][4] - outs[0][4]).max() < 2e-5


@pytest.mark.parametrize("gain,layout", [(1.0, "mocha"), (1.5, "mixamo")])
def test_parity_with_the_oracle(gain, layout):
    """Generator.forward on 48 windows (mid-size tiles of the engine) with the option on against the CPU oracle (fp32 and float64): the
    north-star 1e-4 against the fp32 oracle, and no further from float64 than the default engine is (to 10 %)."""
    from oracle import mocha_oracle as O
    V = 22 if layout == "mixamo" else 24
    sd = weights.synthetic_state_dict(515, gain, layout)
    S = synthetic.pose_windows(41, 48, V); C = synthetic.pose_windows(42, 48, V)
    s32 = O.to_torch_state(sd); s64 = {k: v.double() for k, v in s32.items()}
    with torch.no_grad():
        Y32 = O.generator_forward(s32, torch.from_numpy(S), torch.from_numpy(C)).double()
        Y64 = O.generator_forward(s64, torch.from_numpy(S).double(), torch.from_numpy(C).double())
    model = Generator(layout=layout, device=dev()).load_state_dict(sd).eval()
    res = {}
    for flag in (0, 1):
        model.set_option("gemm_f16x2", flag)
        model.profile_start()
        res[flag] = model(torch.from_numpy(S).to(dev()), torch.from_numpy(C).to(dev())).cpu().double()
        names = set(model.profile_stop()["kernels"])
        assert ("mocha_gemm_h2" in names) == bool(flag), names
    model.set_option("gemm_f16x2", 0)
    scale = max(1.0, float(Y64.abs().max()))
    e = {f: (float((res[f] - Y32).abs().max()), float((res[f] - Y64).abs().max())) for f in (0, 1)}
    print(f"[f16x2] forward, gain {gain}: |hip - oracle32| default {e[0][0]:.2e}, f16x2 {e[1][0]:.2e}; |hip - f64| default {e[0][1]:.2e}, f16x2 {e[1][1]:.2e}; "
          f"|oracle32 - f64| {float((Y32 - Y64).abs().max()):.2e} (max |Y| {scale:.3g})")
    assert e[1][0] < 1e-4 * scale
    assert e[1][1] <= max(e[0][1] * 1.1, 2e-6 * scale)


def test_option_across_the_call_surface():
    """The option on the other routes into the same kernels: chunked batches (three chunks of 64), two streams (two workspace sets, each with
    its own bounds), un-normalised poses (the embedding then stays on the bf16 planes), a bf16-matched bank; repeated calls are
    bit-identical; inputs 1000 x larger / smaller than usual (the bounds, hence the scales, follow - a scale of one would lose them)."""
    V = 22
    sd = weights.synthetic_state_dict(99, 1.0, "mixamo")
    mean, std = synthetic.cnt_norm(7)
    src = torch.from_numpy(synthetic.pose_windows(5, 160, V)).to(dev()); cha = torch.from_numpy(synthetic.pose_windows(6, 160, V)).to(dev())

    def run(setup, f16):
        model = Generator(layout="mixamo", device=dev()).load_state_dict(sd).eval()
        setup(model)
        model.set_option("gemm_f16x2", f16)
        enc, _, nm = model.encode(cha, mean, std)
        bank = ContextBank(model, nm, enc, bf16=True)
        Y1, i1 = bank.characterize(src, mean, std, return_index=True)
        Y2, i2 = bank.characterize(src, mean, std, return_index=True)
        assert torch.equal(Y1, Y2) and torch.equal(i1, i2)
        return Y1.cpu().numpy(), i1.cpu().numpy(), model

    base = run(lambda m: None, 0)
    for name, setup in (("plain", lambda m: None), ("chunks of 64", lambda m: m.reserve(64)),
                        ("two streams", lambda m: (m.set_option("dual_stream", 1), m.set_option("dual_min", 64)))):
        Y, idx, model = run(setup, 1)
        same = idx == base[1]
        assert same.mean() > 0.99 and np.abs(Y[same] - base[0][same]).max() < 2e-5, name
    # scaled inputs through Generator.forward (no matching in between): relative agreement with the default engine at every magnitude
    model = base[2]
    for k in (1e-3, 1.0, 1e3):
        res = {}
        for f16 in (0, 1):
            model.set_option("gemm_f16x2", f16)
            res[f16] = model(src[:64] * k, cha[:64] * k).cpu().double()
        model.set_option("gemm_f16x2", 0)
        d = float((res[1] - res[0]).abs().max()) / max(1e-30, float(res[0].abs().max()))
        print(f"[f16x2] forward on inputs x {k:g}: relative difference to the default engine {d:.2e}")
        assert d < 2e-5, (k, d)
    # un-normalised poses: raw entry points with the option on
    model.set_option("gemm_f16x2", 1)
    nn = (V + 1) * 15
    xm = np.zeros(nn, dtype=np.float32); xs = np.ones(nn, dtype=np.float32)
    model.set_pose_norm(xm, xs, xm, xs)
    raw = torch.cat([torch.zeros((64, 60, 1, 15), device=dev()), src[:64]], 2).contiguous()
    e1, _, _ = model.encode(raw, mean, std, raw=True)
    model.set_option("gemm_f16x2", 0)
    e0, _, _ = model.encode(raw, mean, std, raw=True)
    assert float((e1 - e0).abs().max()) < 2e-5 * max(1.0, float(e0.abs().max()))


def test_few_window_calls_are_untouched_by_the_option():
    """Up to four windows every GEMM of the path runs on the few-rows kernels: with the option on such a call launches neither the engine
    nor its bound kernels / memsets (the streamed per-window step keeps its launch count) and its results are bit-identical."""
    sd = weights.synthetic_state_dict(7, 1.0, "mixamo")
    model = Generator(layout="mixamo", device=dev()).load_state_dict(sd).eval()
    src = torch.from_numpy(synthetic.pose_windows(1, 3, 22)).to(dev()); cha = torch.from_numpy(synthetic.pose_windows(2, 3, 22)).to(dev())
    out = {}
    for f16 in (0, 1):
        model.set_option("gemm_f16x2", f16)
        model.profile_start()
        out[f16] = model(src, cha)
        names = set(model.profile_stop()["kernels"])
        assert "mocha_gemm_h2" not in names and "mocha_absmax" not in names, names
    model.set_option("gemm_f16x2", 0)
    assert torch.equal(out[0], out[1])


def test_a_window_does_not_see_the_rest_of_the_batch():
    """Scales are per window: the poses of the first 24 windows are bit-identical whether the other 72 windows of the batch are ordinary
    ones or 1 000-sigma outliers (a per-batch scale would move them), against a bank and through Generator.forward."""
    V = 22
    sd = weights.synthetic_state_dict(31, 1.0, "mixamo")
    model = Generator(layout="mixamo", device=dev()).load_state_dict(sd).eval()
    model.set_option("gemm_f16x2", 1)
    mean, std = synthetic.cnt_norm(7)
    A = torch.from_numpy(synthetic.pose_windows(1, 24, V)).to(dev())
    B = torch.from_numpy(synthetic.pose_windows(2, 72, V)).to(dev())
    cha = torch.from_numpy(synthetic.pose_windows(3, 96, V)).to(dev())
    enc, _, nm = model.encode(cha, mean, std)
    bank = ContextBank(model, nm, enc)
    Y1, i1 = bank.characterize(torch.cat([A, B]), mean, std, return_index=True)
    Y2, i2 = bank.characterize(torch.cat([A, B * 1000.0]), mean, std, return_index=True)
    assert torch.equal(i1[:24], i2[:24]) and torch.equal(Y1[:24], Y2[:24])
    F1 = model(torch.cat([A, B]), cha); F2 = model(torch.cat([A, B * 1000.0]), torch.cat([cha[:24], cha[24:] * 1000.0]))
    assert torch.equal(F1[:24], F2[:24])
    assert bool(torch.isfinite(Y2).all()) and bool(torch.isfinite(F2).all())
    model.set_option("gemm_f16x2", 0)
