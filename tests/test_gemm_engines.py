"""The two GEMM engines of the library against float64 (run with -m gpu).

* exact-f32 MFMA (gemm_f32.hip, v_mfma_f32_32x32x2_f32) and
* bf16 x 3 planes (gemm_x3.hip: both fp32 operands as three bf16 planes, six v_mfma_f32_32x32x16_bf16 passes, fp32 accumulate),
  the default for the large launches of the path (mocha_set_option("gemm_bf16x3", 1)).

Both replace nn.Linear / conv-as-GEMM of the reference (net/transformer.py:28-32, 57-61; net/blocks.py:57-66, 112-118), whose
arithmetic is fp32.  The claim checked here is that the plane engine is an fp32 GEMM, not a reduced-precision one: its error
against a float64 product is not larger than the exact-f32 MFMA kernel's on the same operands, on the path's shapes, including
operands with a wide dynamic range; and the whole network gives the same poses on either engine (1e-4 is the north-star bound on
the poses; the engines differ by far less)."""
import numpy as np
import pytest
import torch

from mocha_sigasia2023_amd import Generator, synthetic, weights

pytestmark = pytest.mark.gpu


def dev():
    assert torch.cuda.is_available(), "GPU tests need a ROCm device"
    return torch.device("cuda:0")


def _errors(model, x, w, b):
    ref = x.double() @ w.double().T
    if b is not None:
        ref = ref + b.double()
    out = {}
    for name, engine in (("f32", 1), ("x3", 2)):
        y = model.linear(x, w, b, engine=engine)
        d = (y.double() - ref)
        out[name] = (float(d.abs().max()), float(d.pow(2).mean().sqrt()))
    return out, float(ref.abs().max())


@pytest.mark.parametrize("M,N,K,bias", [(105300, 256, 512, True), (105300, 512, 256, False), (30001, 256, 1280, True), (421200, 256, 192, False),
                                        (200000, 128, 32, True), (98305, 384, 64, False),       # the shortest K loops: two and four steps
                                        (210001, 64, 320, True), (105300, 192, 256, False)])     # the 64-wide tile (to_mot's joint block)
def test_plane_engine_is_as_accurate_as_f32_mfma(M, N, K, bias):
    g = torch.Generator(device="cpu").manual_seed(M + N + K)
    x = torch.randn((M, K), generator=g, dtype=torch.float32).to(dev())
    w = (torch.randn((N, K), generator=g, dtype=torch.float32) / np.sqrt(K)).to(dev())
    b = torch.randn((N,), generator=g, dtype=torch.float32).to(dev()) if bias else None
    model = Generator(layout="mixamo", device=dev())
    err, scale = _errors(model, x, w, b)
    # fp32 accumulation of K terms: a few 1e-6 at unit scale; the plane engine must not be worse than the exact-f32 MFMA kernel
    assert err["x3"][1] <= err["f32"][1] * 1.02 + 1e-9, (err, scale)
    assert err["x3"][0] <= err["f32"][0] * 1.5 + 1e-9, (err, scale)
    assert err["f32"][1] < 2e-5 * max(1.0, scale), (err, scale)


def test_plane_engine_wide_dynamic_range():
    """Operands spanning 12 orders of magnitude per row (the planes are exact for every finite fp32 value that is not denormal in
    bf16 range), with heavy cancellation in the sums."""
    M, N, K = 98304, 256, 256
    g = torch.Generator(device="cpu").manual_seed(7)
    mag = torch.pow(10.0, torch.empty((M, K)).uniform_(-6, 6, generator=g))
    x = (torch.randn((M, K), generator=g) * mag).float().to(dev())
    w = (torch.randn((N, K), generator=g) * torch.pow(10.0, torch.empty((N, K)).uniform_(-3, 3, generator=g))).float().to(dev())
    model = Generator(layout="mixamo", device=dev())
    err, scale = _errors(model, x, w, None)
    assert err["x3"][1] <= err["f32"][1] * 1.02, (err, scale)
    assert np.isfinite(err["x3"][0])


def test_engine_selection_and_errors():
    model = Generator(layout="mixamo", device=dev())
    x = torch.randn((256, 64), device=dev())
    w = torch.randn((128, 64), device=dev())
    y0 = model.linear(x, w, None, engine=0)                    # a small launch: the exact-f32 kernels either way
    y1 = model.linear(x, w, None, engine=1)
    assert torch.equal(y0, y1)
    with pytest.raises(RuntimeError, match="outside the bf16x3 engine"):
        model.linear(x, w, None, engine=2)
    with pytest.raises(RuntimeError):
        model.linear(torch.randn((256, 40), device=dev()), torch.randn((128, 40), device=dev()), None)     # K % 32


@pytest.mark.timeout(900)
def test_network_on_either_engine():
    """The demo-pair step at its real size (585 + 585 windows) on both engines: same matches, poses equal to 2e-5."""
    W, V = 585, 22
    sd = weights.synthetic_state_dict(1777, 1.0, "mixamo")
    model = Generator(layout="mixamo", device=dev()).load_state_dict(sd).eval()
    src = torch.from_numpy(synthetic.pose_windows(1777, W, V)).to(dev())
    cha = torch.from_numpy(synthetic.pose_windows(1778, W, V)).to(dev())
    mean, std = synthetic.cnt_norm(7)
    outs = {}
    for flag in (1, 0):
        model.set_option("gemm_bf16x3", flag)
        model.set_option("attention_bf16x3", flag)
        model.profile_start()
        Y, idx = model.characterize_pair(src, cha, mean, std, return_index=True)
        prof_names = set(model.profile_stop()["kernels"])
        assert ("mocha_gemm_x3" in prof_names) == bool(flag), prof_names
        assert ("mocha_attention_x3<128>" in prof_names) == bool(flag) and ("mocha_attention_f32<128>" in prof_names) != bool(flag), prof_names
        outs[flag] = (Y.cpu().numpy(), idx.cpu().numpy())
    model.set_option("gemm_bf16x3", 1)
    model.set_option("attention_bf16x3", 1)
    assert np.array_equal(outs[1][1], outs[0][1]) or (outs[1][1] != outs[0][1]).mean() < 0.01
    same = outs[1][1] == outs[0][1]
    assert np.abs(outs[1][0][same] - outs[0][0][same]).max() < 2e-5


@pytest.mark.parametrize("B", [1, 7, 40])
def test_attention_on_either_engine(B):
    """Encoder and decoder on small batches (exact-f32 GEMMs either way at these sizes): only the attention kernel differs.
    Ragged batches exercise the padded workgroups of the XCD-aware grid."""
    sd = weights.synthetic_state_dict(31, 1.0, "mocha")
    model = Generator(layout="mocha", device=dev()).load_state_dict(sd).eval()
    tok = torch.from_numpy(synthetic.token_features(5, B)).to(dev())
    cha = torch.from_numpy(synthetic.token_features(6, B)).to(dev())
    outs = {}
    for flag in (1, 0):
        model.set_option("attention_bf16x3", flag)
        enc = model.encoder(tok)
        outs[flag] = (enc.cpu().numpy(), model.decoder(enc, cha).cpu().numpy())
    model.set_option("attention_bf16x3", 1)
    for a, b in zip(outs[1], outs[0]):
        scale = max(1.0, float(np.abs(b).max()))
        assert np.abs(a - b).max() < 2e-5 * scale, (np.abs(a - b).max(), scale)


def test_attention_prefetch_variants_are_bit_identical():
    """mocha_attention_x3 has a two-steps-ahead prefetch variant for launches with fewer (window, head) workgroups than CUs
    (B * heads <= 256) and the one-step-ahead variant for full batches: same arithmetic in the same order.  64 windows take the
    first, 65 the second; every GEMM of both calls is the same kernel with per-row arithmetic, so the shared windows must agree
    bit for bit through the encoder (head dim 128) and the decoder (head dim 256)."""
    sd = weights.synthetic_state_dict(31, 1.2)
    model = Generator(device="cuda:0").load_state_dict(sd).eval()
    tok = torch.from_numpy(synthetic.token_features(7, 65)).cuda()
    cha = torch.from_numpy(synthetic.token_features(8, 65)).cuda()
    e65, e64 = model.encoder(tok), model.encoder(tok[:64].contiguous())
    assert torch.equal(e65[:64], e64)
    d65, d64 = model.decoder(tok, cha), model.decoder(tok[:64].contiguous(), cha[:64].contiguous())
    assert torch.equal(d65[:64], d64)


@pytest.mark.parametrize("B", [1, 3, 32])
def test_twelve_wave_decoder_attention_against_the_three_wave_kernel(B):
    """Up to 192 (window, head) pairs (48 windows of four heads) the head-dim-256 attention runs as mocha_attention_x3_split (four wave groups, each a quarter of
    the head dim, partial scores summed through LDS): same products, another association of the score sums.  Against the
    three-wave kernel on the same decoder call the outputs agree to fp32 rounding, and against the exact-f32 attention as closely
    as that kernel does.  B = 3 leaves five of the eight padded workgroups of the XCD-aware grid without a window."""
    sd = weights.synthetic_state_dict(31, 1.2)
    model = Generator(device="cuda:0").load_state_dict(sd).eval()
    tok = torch.from_numpy(synthetic.token_features(7, B)).cuda()
    cha = torch.from_numpy(synthetic.token_features(8, B)).cuda()
    try:
        d_split = model.decoder(tok, cha).clone()
        again = model.decoder(tok, cha)
        assert torch.equal(d_split, again)                       # deterministic: the partial sums are added in group order
        model.set_option("attention_split_max", 0)
        d_three = model.decoder(tok, cha).clone()
        model.set_option("attention_bf16x3", 0)
        d_f32 = model.decoder(tok, cha).clone()
    finally:
        model.set_option("attention_bf16x3", 1)
        model.set_option("attention_split_max", 192)
    scale = max(1.0, float(d_three.abs().max()))
    assert float((d_split - d_three).abs().max()) < 2e-6 * scale
    assert float((d_split - d_f32).abs().max()) < 2e-5 * scale
    assert not torch.equal(d_split, d_three) or B == 0           # it IS another kernel (a silent fallback would be bit-identical)


@pytest.mark.parametrize("B", [49, 64, 200])
def test_decoder_attention_from_key_value_images(B):
    """with the decoder fold the keys IN(cha) and values cha are the same for all heads and layers;
    mocha_instnorm writes them once as pre-split bf16 plane images and mocha_attention_x3_kv (one workgroup per window and head
    pair) reads those.  Same arithmetic as mocha_attention_x3<256> on the fp32 rows (set_option("attention_kv", 0)): equal to
    <= 1e-6 of the output's scale; and both against the exact-f32 MFMA attention.  B = 49 is the first batch past the
    twelve-wave small-batch kernel (49 x 4 = 196 pairs > 192); 200 leaves a ragged last group of eight windows."""
    sd = weights.synthetic_state_dict(21, 2.0)
    model = Generator(device=dev()).load_state_dict(sd).eval()
    src = torch.from_numpy(synthetic.token_features(31, B)).to(dev())
    cha = torch.from_numpy(synthetic.token_features(32, B)).to(dev())
    out = {}
    for name, kv, x3 in (("images", 1, 1), ("rows", 0, 1), ("f32", 0, 0)):
        model.set_option("attention_kv", kv); model.set_option("attention_bf16x3", x3)
        model.profile_start()
        out[name] = model.decoder(src, cha).clone()
        kern = model.profile_stop()["kernels"]
        assert ("mocha_attention_x3_kv<256>" in kern) == (name == "images"), (name, sorted(kern))
    model.set_option("attention_kv", 1); model.set_option("attention_bf16x3", 1)
    scale = float(out["f32"].abs().max())
    e_ir = float((out["images"] - out["rows"]).abs().max()); e_if = float((out["images"] - out["f32"]).abs().max())
    print(f"[attention_kv] B = {B}: |images - rows| = {e_ir:.2e}, |images - exact f32| = {e_if:.2e}, max |out| = {scale:.3g}")
    assert e_ir <= 1e-6 * scale and e_if <= 2e-6 * scale
    # and through the gathered path (cha = bank rows picked by index, read by the instance norm itself)
    mean, std = synthetic.cnt_norm(7)
    X = torch.from_numpy(synthetic.pose_windows(33, B)).to(dev())
    bank_X = torch.from_numpy(synthetic.pose_windows(34, 40)).to(dev())
    e, _, nm = model.encode(bank_X, mean, std)
    from mocha_sigasia2023_amd import ContextBank
    bank = ContextBank(model, nm, e)
    Y1, i1 = bank.characterize(X, mean, std, return_index=True)
    model.set_option("attention_kv", 0)
    Y0, i0 = bank.characterize(X, mean, std, return_index=True)
    model.set_option("attention_kv", 1)
    assert torch.equal(i0, i1) and float((Y1 - Y0).abs().max()) <= 1e-6 * max(1.0, float(Y0.abs().max()))


@pytest.mark.parametrize("persistent", [768, 256, 8])
def test_persistent_plane_gemm_is_bit_identical(persistent):
    """mocha_gemm_x3p runs the K loop ACROSS tiles - a workgroup's last two steps of a tile prepare the next tile's first
    two (no prologue), the epilogue goes through the one LDS stage the last step read from.  Same products in the same order per output
    element: the network's output and every mocha_linear shape (plain / bias; M with ragged last tile; more and fewer tiles than
    workgroups) are bit-identical to mocha_gemm_x3.  `persistent` = workgroups: 768 (three per CU), 256, 8 (every workgroup walks many tiles)."""
    sd = weights.synthetic_state_dict(23, 1.5)
    model = Generator(device=dev()).load_state_dict(sd).eval()
    g = torch.Generator(device=dev()); g.manual_seed(5)
    src = torch.from_numpy(synthetic.pose_windows(41, 70)).to(dev()); cha = torch.from_numpy(synthetic.pose_windows(42, 70)).to(dev())
    shapes = [(6300, 256, 256), (6300, 512, 256), (12345, 256, 512), (1000, 1536, 256), (20000, 128, 128)]
    ops = [(torch.randn((M, K), device=dev(), generator=g), torch.randn((N, K), device=dev(), generator=g), torch.randn((N,), device=dev(), generator=g)) for M, N, K in shapes]
    out = {}
    for pers in (0, persistent):
        model.set_option("gemm_persistent_max_n", 1 << 30)
        model.set_option("gemm_persistent", pers)
        model.profile_start()
        Y = model(src, cha)
        kern = model.profile_stop()["kernels"]
        out[pers] = [Y.clone()] + [model.linear(x, w, b, engine=2).clone() for x, w, b in ops] + [model.linear(x, w, None, engine=2).clone() for x, w, b in ops]
    model.set_option("gemm_persistent", 768); model.set_option("gemm_persistent_max_n", 512)        # the library's defaults (process-wide)
    for a, b in zip(out[0], out[persistent]):
        assert torch.equal(a, b), float((a - b).abs().max())


def test_register_resident_plane_gemm_is_bit_identical():
    """Round 6: mocha_gemm_x3r keeps a wave's 32 activation rows as planes in registers for all the n-tiles of a panel
    and streams only weights (option "gemm_x3r_min_n"; a measured negative inside the step, kept as an option).  Same plane values, same
    products in the same order per output element: mocha_linear shapes (plain / bias; ragged last panel; more and fewer tile pairs than
    workgroups) and the whole network - enc.qkv, dec.q plain, xf.ff1 with bias + GELU - are bit-identical to the tiled instances."""
    sd = weights.synthetic_state_dict(29, 1.5)
    model = Generator(device=dev()).load_state_dict(sd).eval()
    g = torch.Generator(device=dev()); g.manual_seed(6)
    src = torch.from_numpy(synthetic.pose_windows(43, 100)).to(dev()); cha = torch.from_numpy(synthetic.pose_windows(44, 100)).to(dev())
    shapes = [(8320, 512, 256), (9001, 1536, 256), (8200, 1024, 256), (40000, 256, 256), (70000, 512, 256)]
    ops = [(torch.randn((M, K), device=dev(), generator=g), torch.randn((N, K), device=dev(), generator=g), torch.randn((N,), device=dev(), generator=g)) for M, N, K in shapes]
    out, kern = {}, {}
    for min_n in (0, 256):
        model.set_option("gemm_x3r_min_n", min_n)
        model.profile_start()
        Y = model(src, cha)
        kern[min_n] = model.profile_stop()["kernels"]
        out[min_n] = [Y.clone()] + [model.linear(x, w, b, engine=2).clone() for x, w, b in ops] + [model.linear(x, w, None, engine=2).clone() for x, w, b in ops]
    model.set_option("gemm_x3r_min_n", 0)
    assert "mocha_gemm_x3r" in kern[256] and "mocha_gemm_x3r" not in kern[0]
    for a, b in zip(out[0], out[256]):
        assert torch.equal(a, b), float((a - b).abs().max())


@pytest.mark.parametrize("windows,cap", [(1, 512), (2, 512), (7, 3), (37, 512), (150, 64), (300, 512)])
def test_embed_sums_is_the_two_kernel_path(windows, cap):
    """Round 4: mocha_embed_sums_x3 = mocha_embed_front_x3 + mocha_window_sums<48> with the 192-channel frame rows kept in an LDS ring.
    A workgroup takes a contiguous range of pooled frames that may start and end inside a window (every piece pays one extra step and
    reflects at the window's ends): the same per-frame products and the same four additions in the same order, so the embedding is
    bit-identical - for one window, ranges shorter than a window (more workgroups than windows), ranges of many windows (cap = 3 / 64
    workgroups), z-scored raw poses included."""
    sd = weights.synthetic_state_dict(29, 1.5)
    model = Generator(device=dev()).load_state_dict(sd).eval()
    X = torch.from_numpy(synthetic.pose_windows(77 + windows, windows)).to(dev())
    V = X.shape[2]
    g = torch.Generator(device="cpu"); g.manual_seed(windows)
    nn = (V + 1) * 15
    model.set_pose_norm(torch.randn(nn, generator=g).numpy(), (0.5 + torch.rand(nn, generator=g)).numpy(),
                        torch.randn(nn, generator=g).numpy(), (0.5 + torch.rand(nn, generator=g)).numpy())
    Xraw = torch.randn((windows, 60, V + 1, 15), generator=g).to(dev())
    out = {}
    try:
        for fused in (0, 1):
            model.set_option("embed_sums", fused); model.set_option("embed_front_max_wgs", cap)
            model.profile_start()
            tok = model.mot_embedding(X).clone()
            kern = model.profile_stop()["kernels"]
            assert ("mocha_embed_sums_x3" in kern) == bool(fused) and ("mocha_window_sums" in kern) != bool(fused), kern.keys()
            enc, cnt = model.encode(Xraw, raw=True)[:2]
            out[fused] = (tok, enc.clone(), cnt.clone())
    finally:
        model.set_option("embed_sums", 1); model.set_option("embed_front_max_wgs", 512)            # the library's defaults (process-wide)
    for a, b in zip(out[0], out[1]):
        assert torch.equal(a, b), float((a - b).abs().max())
