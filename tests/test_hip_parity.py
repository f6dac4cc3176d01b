"""GPU parity tests (run with -m gpu on an MI355X): the HIP path, called through the C ABI
(ctypes -> libmocha_hip.so), against (a) the golden fixtures produced by the reference itself
and (b) the CPU oracle on seeded inputs.  Tolerance of the north star: outputs within 1e-4
(fp32, per joint and channel)."""
import ast
import os

import numpy as np
import pytest
import torch

from mocha_sigasia2023_amd import ContextBank, Generator, mean_variance_norm, synthetic, weights
from oracle import mocha_oracle as O

pytestmark = pytest.mark.gpu

VARIANTS = ["mocha24_g1", "mocha24_g2", "mixamo22_g1"]
TOL = 1e-4          # absolute on the pose output Y (north star)
RTOL = 1e-4         # intermediates: relative to max(1, max|ref|)

_models = {}


def dev():
    assert torch.cuda.is_available(), "GPU tests need a ROCm device"
    return torch.device("cuda:0")


def load(golden_dir, name):
    z = np.load(os.path.join(golden_dir, f"generator_{name}.npz"))
    meta = ast.literal_eval(str(z["meta"]))
    key = (meta["seed"], meta["gain"], meta["layout"])
    if key not in _models:
        sd = weights.synthetic_state_dict(meta["seed"], meta["gain"], meta["layout"])
        _models[key] = (Generator(layout=meta["layout"], device=dev()).load_state_dict(sd).eval(), sd)
    return z, meta, _models[key][0], _models[key][1]


def rel(a, b):
    a = a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    return float(np.abs(a - b).max() / max(1.0, np.abs(b).max()))


def absmax(a, b):
    a = a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    return float(np.abs(a - b).max())


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev())


@pytest.mark.parametrize("name", VARIANTS)
def test_mot_embedding(golden_dir, name):
    z, meta, model, _ = load(golden_dir, name)
    for tag in ("src", "cha"):
        tokens = model.mot_embedding(T(z[f"{tag}_X"]))
        assert rel(tokens, z[f"{tag}_tokens"]) < RTOL


@pytest.mark.parametrize("name", VARIANTS)
def test_encoder_and_cnt(golden_dir, name):
    z, meta, model, _ = load(golden_dir, name)
    tokens = T(z["src_tokens"])
    tokens = tokens + model.pos_emb[:, :tokens.shape[1]]          # the caller's add, test_fullframework.py:191
    enc = model.encoder(tokens)
    assert rel(enc, z["src_encoded"]) < RTOL
    cnt = mean_variance_norm(T(z["src_encoded"]).permute(0, 2, 1)).permute(0, 2, 1)
    assert rel(cnt, z["src_cnt"]) < RTOL


@pytest.mark.parametrize("name", VARIANTS)
def test_fused_encode(golden_dir, name):
    z, meta, model, _ = load(golden_dir, name)
    enc, cnt = model.encode(T(z["cha_X"]))
    assert rel(enc, z["cha_encoded"]) < RTOL
    assert rel(cnt, z["cha_cnt"]) < RTOL
    mean, std = synthetic.cnt_norm(5)
    enc2, cnt2, nm = model.encode(T(z["cha_X"]), mean, std)
    assert torch.equal(enc, enc2) and torch.equal(cnt, cnt2)      # deterministic, same kernels
    assert rel(nm, O.znorm(z["cha_cnt"], mean, std)) < 3e-4       # division by std/temp_weight amplifies by up to 6


@pytest.mark.parametrize("name", VARIANTS)
def test_decoder(golden_dir, name):
    z, meta, model, _ = load(golden_dir, name)
    dec = model.decoder(T(z["src_encoded"]), T(z["cha_encoded"]))
    assert rel(dec, z["decoded"]) < RTOL


@pytest.mark.parametrize("name", VARIANTS)
def test_to_mot(golden_dir, name):
    z, meta, model, _ = load(golden_dir, name)
    Y = model.to_mot(T(z["decoded"]))
    assert Y.shape == z["Y"].shape
    assert absmax(Y, z["Y"]) < TOL                      # absolute (north star: 1e-4 per joint and channel)


@pytest.mark.parametrize("name", VARIANTS)
def test_generator_forward(golden_dir, name):
    z, meta, model, _ = load(golden_dir, name)
    Y = model(T(z["src_X"]), T(z["cha_X"]))
    assert absmax(Y, z["Y_forward"]) < TOL
    se, ce, sc, cc = model(T(z["src_X"]), T(z["cha_X"]), extract_feature=True)
    assert rel(se, z["src_encoded"]) < RTOL and rel(ce, z["cha_encoded"]) < RTOL
    assert rel(sc, z["src_cnt"]) < RTOL and rel(cc, z["cha_cnt"]) < RTOL


def test_match_against_balltree_fixture(golden_dir):
    z = np.load(os.path.join(golden_dir, "match_balltree.npz"))
    _, _, model, _ = load(golden_dir, "mocha24_g1")
    mean, std = synthetic.cnt_norm(90)
    for nb in (64, 585):
        seed, nb_, q = (int(v) for v in z[f"n{nb}_seed"])
        cha = synthetic.token_features(seed, nb_)
        src = synthetic.token_features(seed + 1, q)
        src[:4] = cha[[3, nb_ - 1, nb_ // 2, 7]] + 0.05 * src[:4]
        bank = ContextBank(model, T(O.znorm(cha, mean, std)), T(cha))
        dist, idx = bank.query(T(O.znorm(src, mean, std)), k=1)
        assert np.array_equal(idx[:, 0].cpu().numpy().astype(np.int64), z[f"n{nb}_idx"])     # bit-exact indices
        assert np.allclose(dist[:, 0].cpu().numpy(), z[f"n{nb}_dist"], rtol=1e-5)
        got = bank.gather(idx)
        assert torch.equal(got.cpu(), torch.from_numpy(cha[z[f"n{nb}_idx"]]))


@pytest.mark.parametrize("B,chunk", [(1, 256), (5, 2), (33, 16)])
def test_characterize_vs_oracle(B, chunk):
    """Whole NN-branch pipeline incl. ragged chunking (B not a multiple of the chunk)."""
    sd = weights.synthetic_state_dict(31, 1.5)
    model = Generator(device=dev()).load_state_dict(sd).eval().reserve(chunk)
    src = synthetic.pose_windows(100 + B, B)
    cha = synthetic.pose_windows(200 + B, 7)
    mean, std = synthetic.cnt_norm(3)
    enc_c, cnt_c, nm_c = model.encode(T(cha), mean, std)
    bank = ContextBank(model, nm_c, enc_c)
    Y, idx = bank.characterize(T(src), mean, std, return_index=True)
    with torch.no_grad():
        Yo, idxo = O.characterize(O.to_torch_state(sd), torch.from_numpy(src), torch.from_numpy(cha), mean, std)
    assert np.array_equal(idx.cpu().numpy(), idxo)
    assert absmax(Y, Yo.numpy()) < TOL


def test_batch_independence_and_determinism():
    """Windows are independent units.  Two runs of the same call agree bit for bit; a
    window's output does not depend on its batch neighbours or the chunking beyond fp32 summation
    order (small batches take the skinny GEMM, which splits K four ways), i.e. far inside 1e-4."""
    sd = weights.synthetic_state_dict(77, 1.0)
    model = Generator(device=dev()).load_state_dict(sd).eval().reserve(8)
    s, c = T(synthetic.pose_windows(1, 9)), T(synthetic.pose_windows(2, 9))
    Y1 = model(s, c)
    Y2 = model(s, c)
    assert torch.equal(Y1, Y2)
    Ys = model(s[4:5].contiguous(), c[4:5].contiguous())
    assert torch.equal(Ys, model(s[4:5].contiguous(), c[4:5].contiguous()))
    assert float((Ys[0] - Y1[4]).abs().max()) < 2e-6 * max(1.0, float(Y1.abs().max()))
    # same batch size, different neighbours: bit-identical (same kernels, same per-row arithmetic)
    s2 = s.clone(); s2[0] = s[8]; c2 = c.clone(); c2[0] = c[8]
    assert torch.equal(model(s2, c2)[4], Y1[4])


def test_errors_are_loud():
    model = Generator(device=dev())
    with pytest.raises(RuntimeError, match="weights not loaded"):
        model.mot_embedding(torch.zeros(1, 60, 24, 15))
    sd = weights.synthetic_state_dict(1, 1.0)
    bad = dict(sd); bad["to_mot.6.bias"] = np.zeros(14, np.float32)
    with pytest.raises(RuntimeError, match="to_mot.6.bias"):
        Generator(device=dev()).load_state_dict(bad)
    with pytest.raises(KeyError):
        Generator(device=dev()).load_state_dict({k: v for k, v in sd.items() if k != "pos_emb"})
    with pytest.raises(ValueError):
        Generator(device=dev()).load_state_dict(sd).mot_embedding(torch.zeros(1, 60, 22, 15))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        Generator(device="cpu")


def _bf16_round(a):
    u = a.view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) >> 16
    return (u.astype(np.uint32) << 16).view(np.float32)


@pytest.mark.parametrize("Q", [1, 2, 3, 8])
@pytest.mark.parametrize("bf16", [False, True])
def test_streaming_matcher(golden_dir, Q, bf16):
    """Few queries / bf16 bank take the HBM-bound bank-scan kernel; indices must equal the exact search
    (for the bf16 bank: the exact search over the bf16-rounded bank)."""
    _, _, model, _ = load(golden_dir, "mocha24_g1")
    mean, std = synthetic.cnt_norm(90)
    nb = 203                                               # ragged: not a multiple of the 16 rows per workgroup
    cha = synthetic.token_features(300, nb)
    src = synthetic.token_features(301, Q)
    src[0] = cha[nb - 1] + 0.05 * src[0]
    cha_nm, src_nm = O.znorm(cha, mean, std), O.znorm(src, mean, std)
    bank = ContextBank(model, T(cha_nm), T(cha), bf16=bf16)
    dist, idx = bank.query(T(src_nm), k=1)
    if bf16:       # the bf16 bank stores bf16(b - centroid); queries are centred the same way (distances are translation invariant)
        c = cha_nm.astype(np.float64).mean(0).astype(np.float32)
        ridx, rdist = O.match_bruteforce(src_nm - c, _bf16_round(np.ascontiguousarray(cha_nm - c)))
    else:
        ridx, rdist = O.match_bruteforce(src_nm, cha_nm)
    assert np.array_equal(idx[:, 0].cpu().numpy().astype(np.int64), ridx)
    assert np.allclose(dist[:, 0].cpu().numpy(), rdist, rtol=1e-5)
    assert idx[0, 0].item() == nb - 1


@pytest.mark.parametrize("Q", [1, 5, 8, 9, 37, 200])
def test_matcher_ranks_near_duplicate_rows_far_from_the_origin(Q):
    """The hard case for an fp32 matcher: one character's bank rows sit close together far from the origin
    (||b||^2 ~ 2e5, nearest-neighbour gaps ~ 1e-2 in d^2), where ||b||^2 - 2 q.b cannot rank in fp32.  The scan evaluates
    (q-b)^2 directly and the GEMM path centres its operands and re-ranks its best candidates exactly: indices must equal the
    float64 brute-force search for every query count (scan: Q <= 8, GEMM: Q > 8)."""
    _, _, model, _ = load(os.path.join(os.path.dirname(__file__), "golden"), "mocha24_g1")
    r = np.random.Generator(np.random.PCG64(77))
    N = 157
    offset = (3.0 * r.standard_normal((1, 90 * 256))).astype(np.float32)
    bank = (offset + 0.02 * r.standard_normal((N, 90 * 256))).astype(np.float32).reshape(N, 90, 256)
    q = (offset + 0.02 * r.standard_normal((Q, 90 * 256))).astype(np.float32).reshape(Q, 90, 256)
    q[0] = bank[N - 1] + 0.001 * r.standard_normal((90, 256)).astype(np.float32)
    ridx, rdist = O.match_bruteforce(q, bank)                      # float64
    b = ContextBank(model, T(bank), T(bank))
    dist, idx = b.query(T(q), k=1)
    assert np.array_equal(idx[:, 0].cpu().numpy().astype(np.int64), ridx)
    assert np.allclose(dist[:, 0].cpu().numpy(), rdist, rtol=2e-4)
    assert idx[0, 0].item() == N - 1


@pytest.mark.parametrize("bf16", [False, True])
def test_matcher_reevaluates_every_row_inside_the_error_bound(bf16):
    """the select kernel used to keep only the 8 best coarse scores when more rows fell inside the coarse
    pass's error bound.  A bank of 40 near-duplicates of each query (d^2 differing in the 4th digit, far below the bf16
    pass's resolution) puts > 8 rows inside the bound; the answer must still be the exact search - float64 over the rows
    the kernel scans (the bf16-rounded centred bank for bf16 banks), ties to the lowest index."""
    _, _, model, _ = load(os.path.join(os.path.dirname(__file__), "golden"), "mocha24_g1")
    r = np.random.Generator(np.random.PCG64(123))
    D = 90 * 256
    Q, dup = 24, 40
    base = r.standard_normal((Q, D)).astype(np.float32)
    rows = (base[:, None, :] + 0.01 * (1.0 + 0.001 * r.standard_normal((Q, dup, 1))) * r.standard_normal((Q, dup, D))).astype(np.float32)
    bank = rows.reshape(Q * dup, D)
    bank = bank[r.permutation(Q * dup)]
    bank[5] = bank[700]                                       # an exact duplicate pair: the lower index must win
    q = base
    b = ContextBank(model, T(bank.reshape(-1, 90, 256)), T(bank.reshape(-1, 90, 256)), bf16=bf16)
    dist, idx = b.query(T(q.reshape(Q, 90, 256)), k=1)
    idx = idx[:, 0].cpu().numpy().astype(np.int64)
    if bf16:
        c = bank.astype(np.float64).mean(0).astype(np.float32)            # the library's centroid is the fp64 column mean, rounded once
        scanned = torch.from_numpy(bank - c).to(torch.bfloat16).to(torch.float64).numpy()
        qq = (q - c).astype(np.float64)
    else:
        scanned = bank.astype(np.float64); qq = q.astype(np.float64)
    d2 = ((qq[:, None, :] - scanned[None, :, :]) ** 2).sum(-1) if Q * scanned.shape[0] * D < 3e8 else None
    if d2 is None:
        d2 = np.stack([((qq[i][None, :] - scanned) ** 2).sum(-1) for i in range(Q)])
    ref = d2.argmin(1)
    # fp32 evaluation of the exact distances: rows whose float64 distances agree to 1e-6 relative may legitimately swap
    for i in range(Q):
        assert idx[i] == ref[i] or abs(d2[i, idx[i]] - d2[i, ref[i]]) <= 2e-6 * d2[i, ref[i]], (i, idx[i], ref[i], d2[i, idx[i]], d2[i, ref[i]])
    assert np.allclose(dist[:, 0].cpu().numpy() ** 2, d2[np.arange(Q), idx], rtol=1e-4)


def test_bf16_bank_many_queries_agrees_with_fp32():
    """BASELINE configs[2] shape in small: bf16 bank, many queries; report-style agreement check."""
    sd = weights.synthetic_state_dict(9, 1.0)
    model = Generator(device=dev()).load_state_dict(sd).eval()
    mean, std = synthetic.cnt_norm(4)
    cha = O.znorm(synthetic.token_features(400, 96), mean, std)
    src = O.znorm(synthetic.token_features(401, 40), mean, std)
    i32 = ContextBank(model, T(cha), T(cha)).query(T(src), return_distance=False)[:, 0].cpu().numpy()
    i16 = ContextBank(model, T(cha), T(cha), bf16=True).query(T(src), return_distance=False)[:, 0].cpu().numpy()
    assert np.array_equal(i32, O.match_bruteforce(src, cha)[0])
    assert (i32 == i16).mean() >= 0.95                    # random N(0,1) banks: gaps >> bf16 rounding


@pytest.mark.parametrize("B", [9, 40, 130])
def test_characterize_on_a_bf16_bank_is_encode_then_query(B):
    """Against a bf16 bank with more than 8 windows, characterize()'s instance norm writes the matcher's centred bf16 query plane
    itself (InormExtra::zc16) instead of a mocha_center_bf16 launch on the z-scored features.  Same subtraction, same rounding: the
    indices must be those of encode() -> ContextBank.query() on the same windows, and the poses those of the decoder on the rows
    those indices gather."""
    sd = weights.synthetic_state_dict(12, 1.3)
    model = Generator(device=dev()).load_state_dict(sd).eval()
    mean, std = synthetic.cnt_norm(6)
    r = np.random.Generator(np.random.PCG64(77))
    src = T(synthetic.pose_windows(31, B))
    with torch.no_grad():
        _, _, nm = model.encode(src, T(mean), T(std))
    nm = nm.reshape(B, -1)
    # a bank with near neighbours for every query (so that a differently rounded query plane WOULD change candidates) and noise rows
    bank_nm = torch.cat([nm + T((2e-3 * r.standard_normal(nm.shape)).astype(np.float32)),
                         nm + T((3e-3 * r.standard_normal(nm.shape)).astype(np.float32)),
                         T(r.standard_normal((300, nm.shape[1])).astype(np.float32))])
    bank_enc = T(r.standard_normal((bank_nm.shape[0], 90, 256)).astype(np.float32))
    bank = ContextBank(model, bank_nm, bank_enc, bf16=True)
    Y, idx = bank.characterize(src, T(mean), T(std), return_index=True)
    iq = bank.query(nm.contiguous(), return_distance=False)[:, 0]
    assert torch.equal(idx.view(-1).to(torch.int64), iq.view(-1).to(torch.int64))
    with torch.no_grad():
        enc, _, _ = model.encode(src, T(mean), T(std))
        Y2 = model.to_mot(model.decoder(enc, bank.gather(iq.to(torch.int32).contiguous())))
    assert float((Y - Y2).abs().max()) < 1e-5


@pytest.mark.parametrize("lanes", [1, 2, 3])
def test_streaming_lanes_reproduce_the_synchronous_steps(lanes):
    """BASELINE configs[4] pipelined: run_clip keeps up to three captured per-window steps in flight on their own lanes (workspace
    set, match scratch, graph, stream each).  Same kernels per window as step(): outputs and indices bit for bit, for every lane
    count, twice in a row (buffers are reused across the clip), and step() still works afterwards."""
    from mocha_sigasia2023_amd import StreamingCharacterizer
    sd = weights.synthetic_state_dict(56, 1.2)
    model = Generator(device=dev()).load_state_dict(sd).eval()
    mean, std = synthetic.cnt_norm(8)
    cha = T(synthetic.pose_windows(72, 41))
    src = T(synthetic.pose_windows(73, 17))
    enc_c, cnt_c, nm_c = model.encode(cha, mean, std)
    bank = ContextBank(model, nm_c, enc_c)
    ref = StreamingCharacterizer(bank, mean, std)
    Ys, Is = [], []
    for i in range(17):
        y, ix = ref.step(src[i])
        Ys.append(y.clone()); Is.append(int(ix.item()))
    sc = StreamingCharacterizer(bank, mean, std, lanes=lanes)
    for _ in range(2):
        Y, idx = sc.run_clip(src)
        torch.cuda.synchronize()
        assert idx.cpu().tolist() == Is
        assert torch.equal(Y, torch.stack(Ys))
    y, ix = sc.step(src[3])
    assert torch.equal(y, Ys[3]) and int(ix.item()) == Is[3]
    # fewer windows than lanes, no windows at all, and a bank switch between two clips (the lanes' graphs are re-captured)
    Y1, i1 = sc.run_clip(src[:1])
    assert torch.equal(Y1[0], Ys[0]) and i1.cpu().tolist() == Is[:1]
    Y0, i0 = sc.run_clip(src[:0])
    assert Y0.shape[0] == 0 and i0.shape[0] == 0
    other = ContextBank(model, nm_c[:7].clone(), enc_c[:7].clone())          # becomes the context's current bank
    assert other.query(nm_c[:3], return_distance=False)[:, 0].cpu().tolist() == [0, 1, 2]
    Y2, i2 = sc.run_clip(src)                                                # ours again: re-activated once, before the lanes fork
    torch.cuda.synchronize()
    assert i2.cpu().tolist() == Is and torch.equal(Y2, torch.stack(Ys))
    model.set_option("lanes", 1)


def test_streaming_graph_replay_matches_batched():
    """BASELINE configs[4] in small: windows streamed one per step through a captured HIP graph must
    reproduce the batched NN-branch result bit for bit (same kernels, same order per window)."""
    from mocha_sigasia2023_amd import StreamingCharacterizer
    sd = weights.synthetic_state_dict(55, 1.2)
    model = Generator(device=dev()).load_state_dict(sd).eval()
    mean, std = synthetic.cnt_norm(8)
    cha = T(synthetic.pose_windows(70, 37))
    src = T(synthetic.pose_windows(71, 6))
    enc_c, cnt_c, nm_c = model.encode(cha, mean, std)
    bank = ContextBank(model, nm_c, enc_c)
    Yb, ib = bank.characterize(src, mean, std, return_index=True)
    with torch.no_grad():
        Yo, io = O.characterize(O.to_torch_state(sd), src.cpu(), cha.cpu(), mean, std)
    assert np.array_equal(ib.cpu().numpy(), io)
    for use_graph in (False, True):
        sc = StreamingCharacterizer(bank, mean, std, use_graph=use_graph)
        for i in range(src.shape[0]):
            y, idx = sc.step(src[i])
            assert int(idx.item()) == int(io[i])
            assert absmax(y, Yo[i].numpy()) < TOL
            # the zero-copy form: the producer writes the window into the step's own input, step() takes no argument
            y1 = y.clone()
            sc.input.zero_(); sc.input.copy_(src[i][None])
            y2, idx2 = sc.step()
            assert torch.equal(y2, y1) and int(idx2.item()) == int(io[i])


def test_fused_pose_normalisation():
    """SURVEY §8 rows a1 + a13: raw poses with the root bone in, de-normalised poses out, against the
    reference's NumPy pre/post-processing (test_fullframework.py:186, 303) around the oracle."""
    sd = weights.synthetic_state_dict(21, 1.3)
    model = Generator(device=dev()).load_state_dict(sd).eval()
    rng = np.random.Generator(np.random.PCG64(5))
    Xm = rng.standard_normal((25, 15)).astype(np.float32); Xs = rng.uniform(0.5, 2.0, (25, 15)).astype(np.float32)
    Ym = rng.standard_normal((25, 15)).astype(np.float32); Ys = rng.uniform(0.5, 2.0, (25, 15)).astype(np.float32)
    model.set_pose_norm(Xm[None, None], Xs[None, None], Ym[None, None], Ys[None, None])   # norm.npz shapes (1,1,25,15)
    src_raw = (rng.standard_normal((4, 60, 25, 15)) * 2 + 1).astype(np.float32)
    cha_raw = (rng.standard_normal((9, 60, 25, 15)) * 2 + 1).astype(np.float32)
    mean, std = synthetic.cnt_norm(6)
    enc_c, cnt_c, nm_c = model.encode(T(cha_raw), mean, std, raw=True)
    bank = ContextBank(model, nm_c, enc_c)
    Y, idx = bank.characterize(T(src_raw), mean, std, return_index=True, raw=True)
    # reference-style host processing around the oracle
    src = (src_raw[:, :, 1:] - Xm[None, None, 1:]) / Xs[None, None, 1:]
    cha = (cha_raw[:, :, 1:] - Xm[None, None, 1:]) / Xs[None, None, 1:]
    with torch.no_grad():
        Yo, io = O.characterize(O.to_torch_state(sd), torch.from_numpy(src), torch.from_numpy(cha), mean, std)
    Yo = Yo.numpy() * Ys[None, None, 1:] + Ym[None, None, 1:]
    assert np.array_equal(idx.cpu().numpy(), io)
    assert absmax(Y, Yo) < TOL * max(1.0, np.abs(Yo).max())
    with pytest.raises(RuntimeError, match="set_pose_norm"):
        Generator(device=dev()).load_state_dict(sd).encode(T(cha_raw), raw=True)


def test_dual_stream_option_matches_single_stream():
    """mocha_set_option("dual_stream", 1): the two halves of a large batch run on two streams; per window the result may
    only differ by the kernel choice of a half-size batch (fp32 summation order), and a graph capture must still work."""
    sd = weights.synthetic_state_dict(5, 1.3)
    model = Generator(device=dev()).load_state_dict(sd).eval()
    mean, std = synthetic.cnt_norm(2)
    src, cha = T(synthetic.pose_windows(3, 150)), T(synthetic.pose_windows(4, 40))
    enc_c, _, nm_c = model.encode(cha, mean, std)
    bank = ContextBank(model, nm_c.clone(), enc_c.clone())
    Y1, i1 = bank.characterize(src, mean, std, return_index=True)
    model.set_option("dual_stream", 1).set_option("dual_min", 64)
    Y2, i2 = bank.characterize(src, mean, std, return_index=True)
    Y3 = model(src[:130].contiguous(), src[20:150].contiguous())
    for _ in range(10):                                          # the two halves overlap on the CUs: still the same bits every time
        Yr, ir = bank.characterize(src, mean, std, return_index=True)
        assert torch.equal(Yr, Y2) and torch.equal(ir, i2)
    model.set_option("dual_stream", 0)
    Y4 = model(src[:130].contiguous(), src[20:150].contiguous())
    assert torch.equal(i1, i2)
    assert float((Y1 - Y2).abs().max()) < 2e-6 * max(1.0, float(Y1.abs().max()))
    assert float((Y3 - Y4).abs().max()) < 2e-6 * max(1.0, float(Y4.abs().max()))


@pytest.mark.parametrize("name", VARIANTS)
def test_decoder_projection_folding_is_only_rounding(golden_dir, name):
    """Default path: key / value projections folded into the query / output weights (mocha_set_option "fold_decoder").
    Both the folded and the literal four-projection decoder must meet the fixture tolerance, and agree with each other
    to fp32 rounding."""
    z, meta, model, _ = load(golden_dir, name)
    folded = model.decoder(T(z["src_encoded"]), T(z["cha_encoded"]))
    model.set_option("fold_decoder", 0)
    literal = model.decoder(T(z["src_encoded"]), T(z["cha_encoded"]))
    model.set_option("fold_decoder", 1)
    assert rel(folded, z["decoded"]) < RTOL and rel(literal, z["decoded"]) < RTOL
    assert not torch.equal(folded, literal)                      # the option really switches the kernel chain
    assert float((folded - literal).abs().max()) < 1e-5 * float(literal.abs().max())


@pytest.mark.parametrize("name", VARIANTS)
def test_joint_block_folding_is_only_rounding(golden_dir, name):
    """Default path: the embedding joint block's 1x1 gcn conv folded into its k=5 temporal conv (net/blocks.py:126-134 applies
    them back to back; mocha_set_option "fold_joint").  Folded and literal two-GEMM forms both meet the fixture tolerance and
    agree with each other to fp32 rounding, at fixture size and on a batch large enough for the plane engine."""
    z, meta, model, _ = load(golden_dir, name)
    X = T(z["src_X"])
    folded = model.mot_embedding(X)
    model.set_option("fold_joint", 0)
    literal = model.mot_embedding(X)
    model.set_option("fold_joint", 1)
    assert rel(folded, z["src_tokens"]) < RTOL and rel(literal, z["src_tokens"]) < RTOL
    assert not torch.equal(folded, literal)
    assert float((folded - literal).abs().max()) < 1e-5 * float(literal.abs().max())
    big = T(synthetic.pose_windows(17, 96, X.shape[2]))
    f2 = model.mot_embedding(big)
    model.set_option("fold_joint", 0)
    l2 = model.mot_embedding(big)
    model.set_option("fold_joint", 1)
    assert float((f2 - l2).abs().max()) < 1e-5 * float(l2.abs().max())


@pytest.mark.parametrize("name", VARIANTS)
def test_upsample_folding_is_only_rounding(golden_dir, name):
    """Default path (round 4): to_mot's k = 5 temporal conv over the nearest-x4-upsampled frames (model.py:74, net/blocks.py:112-118) runs
    as a 3-tap conv over the 15 SOURCE frames with per-phase summed weights (output frame 4 s + phase only ever reads source frames
    s-1, s, s+1; the reflection at the upsampled ends stays inside the first / last source frame); mocha_set_option "fold_upsample".
    Folded and literal forms both meet the fixture tolerance and agree to fp32 rounding - at fixture size (fp32 kernels), on a batch
    large enough for the plane engine, with both engines, and on every frame (the first and last source frames are where the
    reflection matters)."""
    z, meta, model, _ = load(golden_dir, name)
    dec = T(z["decoded"])
    folded = model.to_mot(dec)
    model.set_option("fold_upsample", 0)
    literal = model.to_mot(dec)
    model.set_option("fold_upsample", 1)
    assert absmax(folded, z["Y"]) < TOL and absmax(literal, z["Y"]) < TOL
    assert not torch.equal(folded, literal)
    assert float((folded - literal).abs().max()) < 1e-5 * float(literal.abs().max())
    g = torch.Generator(device="cpu"); g.manual_seed(3)
    for engine in (1, 0):
        model.set_option("gemm_bf16x3", engine)
        for B in (96, 7):
            big = (2.0 * torch.randn((B, 90, 256), generator=g)).to(dev())
            f2 = model.to_mot(big)
            model.set_option("fold_upsample", 0)
            l2 = model.to_mot(big)
            model.set_option("fold_upsample", 1)
            d = (f2 - l2).abs().amax(dim=(0, 2, 3))                 # per frame
            assert float(d.max()) < 1e-5 * float(l2.abs().max()), d
    model.set_option("gemm_bf16x3", 1)


def test_characterize_pair_matches_the_three_call_path():
    """mocha_characterize_pair = encode(cha) + bank_set + characterize(src) with shared launches: same indices, outputs equal
    up to the kernel choice of a larger batch (<= 2e-6 relative), the context's own bank untouched, oracle parity."""
    sd = weights.synthetic_state_dict(21, 1.2)
    model = Generator(device=dev()).load_state_dict(sd).eval()
    mean, std = synthetic.cnt_norm(4)
    src, cha = T(synthetic.pose_windows(5, 37)), T(synthetic.pose_windows(6, 53))
    enc_c, _, nm_c = model.encode(cha, mean, std)
    bank = ContextBank(model, nm_c.clone(), enc_c.clone())
    Y1, i1 = bank.characterize(src, mean, std, return_index=True)
    other = ContextBank(model, nm_c[:5].clone(), enc_c[:5].clone())              # the context's own bank before the pair call
    other.activate()
    Y2, i2, enc2, nm2 = model.characterize_pair(src, cha, mean, std, return_index=True, return_bank=True)
    assert torch.equal(i1, i2)
    scale = max(1.0, float(Y1.abs().max()))
    assert float((Y1 - Y2).abs().max()) < 2e-6 * scale
    assert float((enc2 - enc_c).abs().max()) < 3e-6 * float(enc_c.abs().max())
    assert float((nm2 - nm_c).abs().max()) < 3e-5 * float(nm_c.abs().max())
    d, i = other.query(nm_c[:5], k=1)                                            # still the 5-entry bank
    assert i[:, 0].tolist() == [0, 1, 2, 3, 4]
    # ... with its own centroid and norms: a many-query search (GEMM path on centred operands) must still be exact
    far = ContextBank(model, (nm_c[:40] + 50.0).contiguous(), enc_c[:40].clone())
    far.activate()
    model.characterize_pair(src, cha, mean, std)
    qs_far = (nm_c[:40] + 50.0 + 1e-3 * torch.randn_like(nm_c[:40])).contiguous()
    assert far.query(qs_far, return_distance=False)[:, 0].tolist() == list(range(40))
    ost = O.to_torch_state(sd)
    with torch.no_grad():
        Yo, io = O.characterize(ost, src.cpu(), cha.cpu(), mean, std)
        # random windows give near-equidistant bank rows: where the arg-min differs from the oracle's, it must be a tie to fp32 rounding
        qs = O.znorm(O.encode(ost, src.cpu())[1].numpy(), mean, std).reshape(len(src), -1).astype(np.float64)
        ks = O.znorm(O.encode(ost, cha.cpu())[1].numpy(), mean, std).reshape(len(cha), -1).astype(np.float64)
    ours = i2.cpu().numpy()
    d_ours = np.sqrt(((qs - ks[ours]) ** 2).sum(1)); d_best = np.sqrt(((qs - ks[io]) ** 2).sum(1))
    assert np.all(d_ours <= d_best * (1 + 1e-9))
    same = ours == io
    assert same.all()
    assert absmax(Y2[torch.from_numpy(same).to(Y2.device)], Yo.numpy()[same]) < TOL
    with pytest.raises(RuntimeError, match="workspace limit"):
        model.characterize_pair(T(synthetic.pose_windows(7, 700)), T(synthetic.pose_windows(8, 700)), mean, std)
