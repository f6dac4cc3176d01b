"""Structured-input parity (run with -m gpu): the HIP path against the CPU oracle on inputs that are NOT i.i.d. N(0, 1).

Every other parity test draws its poses from synthetic.pose_windows (white noise).  Real inputs are nothing like that
(VERDICT r3, weak 1): a clip slid with step 1 gives windows that overlap by 59 frames (test_fullframework.py:128), so the bank
is a string of near-duplicates; a standing character gives 60 identical frames; some feature channels are constant; a glitch
in a capture is a 30-50 sigma spike; and a character can be matched against itself.  The two instance norms divide by
`std + 1e-5` (net/transformer.py:13-20), which amplifies rounding wherever a channel's variance over the 90 tokens is small.

Tolerance (north star): |Y - Y_oracle| < 1e-4 absolute, nearest-neighbour indices equal with ties judged in float64 on the
oracle's own features.  Where the fp32 arithmetic itself cannot deliver 1e-4 - the oracle run in float64 on the same inputs says
how far the fp32 ORACLE is from the exact result, and two fp32 runs of the reference (another batch size, another thread count) differ
from EACH OTHER by up to 2e-3 on such inputs (profiles/r05/a_reference_self_consistency.txt) - the HIP path is held to float64 instead:
|hip - f64| <= max(1e-4 max(1, max|Y|), 2 |oracle32 - f64|), the per-row bound the round-5 matrix supports (tests/test_structured_matrix.py,
profiles/r05/a_structured_matrix.txt: worst ratio 1.49 over 240 rows; round 4's factor was 4 and its arithmetic reached 5.0).  The test prints
all three distances (pytest -s) so the numbers land in profiles/.  Why the decoder is ill-conditioned here: AdaIN's gain 1 + gamma is near
zero in some channel, and the instance norm that follows divides every earlier rounding by it (tools/precision_study.py); round 5 evaluates
that pair in closed form and the style MLP in float64, which is why the HIP path is now CLOSER to float64 than the fp32 reference arithmetic
in most rows.
"""
import numpy as np
import pytest
import torch

from mocha_sigasia2023_amd import ContextBank, Generator, StreamingCharacterizer, mean_variance_norm, synthetic, weights
from mocha_sigasia2023_amd.skeleton import LAYOUTS
from oracle import featurize_oracle as FO
from oracle import mocha_oracle as O

pytestmark = pytest.mark.gpu
TOL = 1e-4
DIM = 90 * 256


def dev():
    assert torch.cuda.is_available(), "GPU tests need a ROCm device"
    return torch.device("cuda:0")


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev())


@pytest.fixture(scope="module")
def sd():
    # gain 2: activations and outputs of trained-model size (|Y| up to a few units, like the g2 fixture), so that 1e-4 ABSOLUTE
    # discriminates - at gain 1 the synthetic network's output barely moves with its input (|Y| ~ 0.15, every case within 2e-7)
    return weights.synthetic_state_dict(4242, 2.0)


@pytest.fixture(scope="module")
def model(sd):
    return Generator(device=dev()).load_state_dict(sd).eval()


def _states(sd):
    s32 = O.to_torch_state(sd)
    return s32, {k: v.double() for k, v in s32.items()}


def _encode_batched(st, X, batch=32):
    e, c = zip(*(O.encode(st, X[s:s + batch]) for s in range(0, len(X), batch)))
    return torch.cat(e), torch.cat(c)


def _decode_batched(st, se, sel, batch=32):
    return torch.cat([O.to_mot(st, O.decoder(st, se[s:s + batch], sel[s:s + batch])) for s in range(0, len(se), batch)])


def _ties_ok(ours, best, q64, k64, rtol=1e-6):
    """Every index that differs from the oracle's must be as near as the oracle's winner up to `rtol` (float64, oracle features)."""
    for i in np.nonzero(ours != best)[0]:
        d_o = np.sqrt(((q64[i] - k64[ours[i]]) ** 2).sum())
        d_b = np.sqrt(((q64[i] - k64[best[i]]) ** 2).sum())
        if not d_o <= d_b * (1 + rtol) + 1e-12:
            return False, int(i)
    return True, int((ours != best).sum())


def _check_Y(name, Y_hip, Y32, Y64):
    """|Y - oracle| < 1e-4 absolute; if the fp32 oracle itself is further than that from the float64 result, the HIP path may be up
    to twice as far from float64 as the fp32 oracle is (and must still be finite)."""
    Yh = Y_hip.detach().cpu().numpy().astype(np.float64)
    e_ho = float(np.abs(Yh - Y32.numpy()).max())
    e_h64 = float(np.abs(Yh - Y64.numpy()).max())
    e_o64 = float(np.abs(Y32.numpy().astype(np.float64) - Y64.numpy()).max())
    scale = max(1.0, float(np.abs(Y64.numpy()).max()))
    print(f"[structured] {name}: max|Y| = {scale:.3g}   |hip - oracle32| = {e_ho:.2e}   "
          f"|hip - f64| = {e_h64:.2e}   |oracle32 - f64| = {e_o64:.2e}")
    assert np.isfinite(Yh).all(), name
    assert e_ho < TOL or e_h64 <= max(TOL * scale, 2.0 * e_o64), f"{name}: |hip - oracle| = {e_ho:.3e}, |hip - f64| = {e_h64:.3e}, |oracle - f64| = {e_o64:.3e}"
    return e_ho, e_h64, e_o64


def _forward_case(name, model, sd, src, cha):
    """Generator.forward(src, cha) (model.py:82-106) on both sides, fp32 oracle and float64 oracle."""
    s32, s64 = _states(sd)
    Y = model(T(src), T(cha))
    torch.cuda.synchronize()
    with torch.no_grad():
        Y32 = O.generator_forward(s32, torch.from_numpy(src), torch.from_numpy(cha))
        Y64 = O.generator_forward(s64, torch.from_numpy(src).double(), torch.from_numpy(cha).double())
    return _check_Y(name, Y, Y32, Y64)


# ---------------------------------------------------------------------------------------------------------------- static pose
def test_static_pose_sixty_identical_frames(model, sd):
    """A character standing still: every window is one pose repeated 60 times (both clips).  The temporal convolutions see a
    constant signal through their reflect padding, the 4-frame pool averages identical values, and the tokens differ only by
    body part and position embedding."""
    r = np.random.Generator(np.random.PCG64(101))
    src = np.repeat(r.standard_normal((24, 1, 24, 15)).astype(np.float32), 60, axis=1)
    cha = np.repeat(r.standard_normal((24, 1, 24, 15)).astype(np.float32), 60, axis=1)
    _forward_case("static pose, forward", model, sd, src, cha)
    # the same through the matcher: a bank of static poses, queries = the same poses with a 1e-3 perturbation -> row i
    mean, std = synthetic.cnt_norm(7)
    near = (cha + 1e-3 * r.standard_normal(cha.shape)).astype(np.float32)
    e, _, nm = model.encode(T(cha), mean, std)
    Y, idx = ContextBank(model, nm, e).characterize(T(near), mean, std, return_index=True)
    s32, s64 = _states(sd)
    with torch.no_grad():
        Y32, io = O.characterize(s32, torch.from_numpy(near), torch.from_numpy(cha), mean, std)
        Y64, i64 = O.characterize(s64, torch.from_numpy(near).double(), torch.from_numpy(cha).double(), mean.astype(np.float64), std.astype(np.float64))
    assert np.array_equal(idx.cpu().numpy(), io) and np.array_equal(io, np.arange(24)) and np.array_equal(i64, io)
    # a static pose is the low-variance case of the decoder's instance norms: at these weights the fp32 ORACLE is ~7e-4 from float64
    _check_Y("static pose, characterize", Y, Y32, Y64)


# ------------------------------------------------------------------------------------------------- constant / zeroed channels
def test_constant_and_zeroed_channels(model, sd):
    """Zeroed channel groups (no velocities: channels 9-14 are 0 for every joint), two joints frozen over all windows and frames,
    and one joint whose every channel is the same constant."""
    r = np.random.Generator(np.random.PCG64(102))
    def make(n):
        X = r.standard_normal((n, 60, 24, 15)).astype(np.float32)
        X[..., 9:15] = 0.0
        X[:, :, 3] = r.standard_normal((15,)).astype(np.float32)
        X[:, :, 7] = r.standard_normal((15,)).astype(np.float32)
        X[:, :, 11] = 0.75
        return X
    _forward_case("constant / zeroed channels, forward", model, sd, make(40), make(40))
    _forward_case("all-zero source clip", model, sd, np.zeros((8, 60, 24, 15), np.float32), make(8))


# ------------------------------------------------------------------------------------------------------------------- outliers
def test_thirty_to_fifty_sigma_outliers(model, sd):
    """Capture glitches: a dozen entries per window at +-30 ... 50 sigma on top of N(0, 1) poses, in the source, in the character,
    and in both.  Activations grow with the spikes, and so does every fp32 rounding error: the oracle's own distance from the
    float64 result is printed beside the HIP path's."""
    r = np.random.Generator(np.random.PCG64(103))
    def spiky(n):
        X = r.standard_normal((n, 60, 24, 15)).astype(np.float32)
        for b in range(n):
            k = r.integers(0, 60 * 24 * 15, 12)
            X[b].reshape(-1)[k] = (r.uniform(30, 50, 12) * r.choice([-1.0, 1.0], 12)).astype(np.float32)
        return X
    clean = r.standard_normal((32, 60, 24, 15)).astype(np.float32)
    _forward_case("outliers in the source", model, sd, spiky(32), clean)
    _forward_case("outliers in the character", model, sd, clean, spiky(32))
    _forward_case("outliers in both", model, sd, spiky(32), spiky(32))


# ------------------------------------------------------------------------------------------------------------------ cha == src
def test_character_equals_source(model, sd):
    """forward(X, X) and a clip characterized against its own bank: every window must match itself (distance 0) and the decoder's
    cross-attention sees keys = IN(its own queries' source)."""
    X = synthetic.pose_windows(104, 48)
    _forward_case("cha == src, forward", model, sd, X, X)
    mean, std = synthetic.cnt_norm(7)
    Y, idx, enc, nm = model.characterize_pair(T(X), T(X), mean, std, return_index=True, return_bank=True)
    assert np.array_equal(idx.cpu().numpy(), np.arange(48))
    d, i = ContextBank(model, nm, enc).query(nm)
    assert np.array_equal(i[:, 0].cpu().numpy(), np.arange(48)) and float(d.max()) == 0.0
    s32, s64 = _states(sd)
    with torch.no_grad():
        Y32, io = O.characterize(s32, torch.from_numpy(X), torch.from_numpy(X), mean, std)
        Y64, i64 = O.characterize(s64, torch.from_numpy(X).double(), torch.from_numpy(X).double(), mean.astype(np.float64), std.astype(np.float64))
    assert np.array_equal(io, np.arange(48)) and np.array_equal(i64, io)
    _check_Y("cha == src, characterize", Y, Y32, Y64)


# ------------------------------------------------------------------------------------ the instance norm in low-variance channels
def test_instance_norm_low_variance_channels():
    """mean_variance_norm (net/transformer.py:13-20) where `std + 1e-5` amplifies: constant channels (exact result 0; any rounding
    of the mean is divided by 1e-5), channels whose spread is 1e-4 ... 1e-6 of a large mean, and ordinary ones beside them.  The
    exact result is computed in float64 from the same fp32 inputs.  Bound per channel: a few ulps of |mean| divided by
    (std + 1e-5) - what ANY fp32 evaluation of x - mean(x) can be off by - or 1e-4 where that is smaller.  The fp32 torch
    oracle's own error is printed beside ours."""
    r = np.random.Generator(np.random.PCG64(105))
    B = 6
    x = r.standard_normal((B, 256, 90)).astype(np.float32)                                   # (B, C, S) as the reference passes it
    x[:, 0:16] = r.standard_normal((B, 16, 1)).astype(np.float32)                            # constant over the tokens
    x[:, 16:32] = 0.0
    for j, (mu, sg) in enumerate(((100.0, 1e-3), (100.0, 1e-4), (1000.0, 1e-3), (10.0, 1e-5), (1.0, 1e-6), (-37.5, 3e-4))):
        x[:, 32 + 8 * j: 40 + 8 * j] = (mu + sg * r.standard_normal((B, 8, 90))).astype(np.float32)
    ours = mean_variance_norm(T(x)).cpu().numpy().astype(np.float64)
    x64 = x.astype(np.float64)
    m64 = x64.mean(-1, keepdims=True); s64 = x64.std(-1, ddof=1, keepdims=True)
    ref = (x64 - m64) / (s64 + 1e-5)
    orc = O.mean_variance_norm(torch.from_numpy(x)).numpy().astype(np.float64)
    ulp = np.spacing(np.abs(m64).astype(np.float32)).astype(np.float64)
    bound = np.maximum(TOL, 4.0 * ulp / (s64 + 1e-5) + 4e-7 * np.abs(ref).max(-1, keepdims=True))
    e_h = np.abs(ours - ref).max(-1, keepdims=True); e_o = np.abs(orc - ref).max(-1, keepdims=True)
    print(f"[structured] instance norm: ordinary channels |hip - f64| = {e_h[:, 80:].max():.2e} (oracle {e_o[:, 80:].max():.2e}); "
          f"constant channels {e_h[:, :32].max():.2e} (oracle {e_o[:, :32].max():.2e}); "
          f"low-variance channels {e_h[:, 32:80].max():.2e} (oracle {e_o[:, 32:80].max():.2e}), bound up to {bound[:, 32:80].max():.2e}")
    assert np.isfinite(ours).all()
    assert (e_h <= bound).all(), f"worst excess {float((e_h / bound).max()):.2f} x the bound"
    assert e_h[:, 80:].max() < 1e-5                                                          # ordinary channels: plain fp32 accuracy
    assert np.all(ours[:, 16:32] == 0.0)                                                     # all-zero channels give exactly 0


# ------------------------------------------------------------- a smooth clip, slid with step 1, as source AND as bank (585 windows)
@pytest.mark.timeout(3000)
def test_smooth_clip_stride1_windows_as_source_and_bank(model, sd):
    """The demo's actual shape (test_fullframework.py:124-194, 271-298, 438-443, 465-467): a 644-frame clip -> 585 windows slid
    with step 1 (neighbours overlap by 59 frames) -> featurise on the device -> z-score with norm.npz-style statistics of the clip
    itself (data_loader.py:108-127: std + 1e-6) -> encode -> bank.  (a) the clip against its own bank: every window must pick
    ITSELF among neighbours that differ by one frame; (b) a second take of the same motion (a third of a frame later, 3 % wider
    swing) against that bank: indices equal to the float64 search on the oracle's features, poses within 1e-4; (c) the first
    windows of (b) streamed one at a time through the captured step against the same bank."""
    parents = FO.full_parents(LAYOUTS["mocha"]["parents"])
    clip_a = [synthetic.slide_windows(a) for a in synthetic.smooth_bone_clip(3)]
    clip_b = [synthetic.slide_windows(a) for a in synthetic.smooth_bone_clip(3, phase=0.37, gain=1.03)]
    W = clip_a[0].shape[0]
    assert W == 585
    Xa_o = FO.featurize(*clip_a, parents)
    Xb_o = FO.featurize(*clip_b, parents)
    Xa = model.featurize(*(T(a) for a in clip_a))
    Xb = model.featurize(*(T(a) for a in clip_b))
    scale = max(1.0, float(np.abs(Xa_o).max()))
    e_feat = max(float(np.abs(Xa.cpu().numpy() - Xa_o).max()), float(np.abs(Xb.cpu().numpy() - Xb_o).max()))
    print(f"[structured] smooth clip: featurize |hip - oracle| = {e_feat:.2e} (max |X| = {scale:.3g})")
    assert e_feat < TOL * scale
    # norm.npz of this 'dataset' (data_loader.py:108-127); Y statistics are not on this path's input side
    X_mean = Xa_o.mean(axis=(0, 1)).astype(np.float32)
    X_std = (Xa_o.std(axis=(0, 1)).astype(np.float32) + 1e-6).astype(np.float32)
    ones, zeros = np.ones_like(X_std), np.zeros_like(X_mean)
    model.set_pose_norm(X_mean, X_std, zeros, ones)
    # both sides start from the SAME featurised windows (the device's), so that the comparison below is of the network and the matcher
    Xa_h, Xb_h = Xa.cpu().numpy(), Xb.cpu().numpy()
    za = ((Xa_h[:, :, 1:] - X_mean[None, None, 1:]) / X_std[None, None, 1:]).astype(np.float32)          # test_fullframework.py:186
    zb = ((Xb_h[:, :, 1:] - X_mean[None, None, 1:]) / X_std[None, None, 1:]).astype(np.float32)
    print(f"[structured] smooth clip: z-scored input range [{za.min():.3g}, {za.max():.3g}], smallest X_std {X_std[1:].min():.2e}")
    # cnt_norm.npz of this clip (compute_cnt_norm.py:174-179, std / temporal weight :89): from the oracle's cnt features
    s32, _ = _states(sd)
    with torch.no_grad():
        ea, ca = _encode_batched(s32, torch.from_numpy(za))
        eb, cb = _encode_batched(s32, torch.from_numpy(zb))
    cnt_mean = ca.numpy().mean(0).astype(np.float32)
    cnt_std = ((ca.numpy().std(0) + 1e-6) / synthetic.temporal_weight(15, 6, 256)).astype(np.float32)

    # ---- (a) the clip against its own bank
    Ya, ia, enc_a, nm_a = model.characterize_pair(Xa, Xa, cnt_mean, cnt_std, return_index=True, return_bank=True, raw=True)
    assert np.array_equal(ia.cpu().numpy(), np.arange(W)), "a window of the clip did not match itself in its own bank"
    bank = ContextBank(model, nm_a, enc_a)
    d_self, i_self = bank.query(nm_a)
    assert np.array_equal(i_self[:, 0].cpu().numpy(), np.arange(W)) and float(d_self.max()) == 0.0
    # how near the neighbours are: distance to the next window relative to the distance to a far one
    k64 = O.znorm(ca.numpy(), cnt_mean, cnt_std).reshape(W, -1).astype(np.float64)
    d_next = np.sqrt(((k64[1:] - k64[:-1]) ** 2).sum(1)); d_far = np.sqrt(((k64[W // 2:] - k64[: W - W // 2]) ** 2).sum(1))
    print(f"[structured] smooth clip: bank neighbours {d_next.min():.3g} ... {d_next.max():.3g} apart, half a clip away {d_far.mean():.3g}")
    e_enc = float(np.abs(enc_a.cpu().numpy() - ea.numpy()).max())
    assert e_enc < TOL * max(1.0, float(ea.abs().max())), f"bank features: {e_enc:.3e}"

    # ---- (b) the second take against that bank
    Yb, ib = bank.characterize(Xb, cnt_mean, cnt_std, return_index=True, raw=True)
    torch.cuda.synchronize()
    q64 = O.znorm(cb.numpy(), cnt_mean, cnt_std).reshape(W, -1).astype(np.float64)
    io, _ = O.match_bruteforce(q64, k64)
    ours = ib.cpu().numpy().astype(np.int64)
    ok, info = _ties_ok(ours, io, q64, k64)
    assert ok, f"query {info} matched a row that is not a nearest neighbour"
    same = ours == io
    print(f"[structured] smooth clip: {int(same.sum())} / {W} indices equal the oracle's ({len(np.unique(io))} distinct rows matched)")
    assert same.mean() > 0.99
    _, s64 = _states(sd)
    with torch.no_grad():
        Y32 = _decode_batched(s32, eb, ea[torch.from_numpy(io)])
        ea64, _ = _encode_batched(s64, torch.from_numpy(za).double())
        eb64, _ = _encode_batched(s64, torch.from_numpy(zb).double())
        Y64 = _decode_batched(s64, eb64, ea64[torch.from_numpy(io)])
    pick = torch.from_numpy(np.nonzero(same)[0])
    # smooth motion = slowly varying tokens = low-variance channels in the decoder's instance norms: the float64 criterion applies (header)
    _check_Y(f"smooth clip, {int(same.sum())} windows", Yb.cpu()[pick], Y32[pick], Y64[pick])
    assert torch.isfinite(Yb).all()

    # ---- (c) streamed: one window at a time through the captured step, against the same bank
    sc = StreamingCharacterizer(bank, cnt_mean, cnt_std, use_graph=True, raw=True)
    for w in (0, 1, 2, 100, 291, 584):
        y1, i1 = sc.step(Xb[w])
        assert int(i1.item()) == int(ours[w])
        assert float((y1 - Yb[w]).abs().max()) < 1e-5 * max(1.0, float(Yb.abs().max()))      # one-window kernels: another fp32 summation order
