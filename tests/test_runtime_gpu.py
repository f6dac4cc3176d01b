"""Runtime behaviour of the context through the C ABI (run with -m gpu): introspection entry points, the generation
counter that guards captured graphs, the captured per-window step, the RCCL communicator with one rank, bank-sharded
matching with one rank, repeated weight loads, non-finite inputs, and bench.py's own N-rank launcher."""
import ctypes as C
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from mocha_sigasia2023_amd import ContextBank, Generator, StreamingCharacterizer, synthetic, weights
from mocha_sigasia2023_amd.skeleton import skeleton_constants

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def model():
    return Generator(device="cuda:0").load_state_dict(weights.synthetic_state_dict(11, 1.0)).eval()


def _norm():
    return synthetic.cnt_norm(3)


@pytest.mark.parametrize("layout", ["mocha", "mixamo"])
def test_graph_constants_and_pos_emb_entry_points(layout):
    """mocha_graph_constants regenerates a14's buffers (net/graph.py) in C++; mocha_pos_emb exposes model.pos_emb (model.py:40)."""
    sd = weights.synthetic_state_dict(3, 1.0, layout)
    m = Generator(layout=layout, device="cuda:0").load_state_dict(sd).eval()
    sk = skeleton_constants(layout)
    V = sk.V
    A_j = np.empty((3, V, V), np.float32); A_b = np.empty((2, 6, 6), np.float32)
    pool = np.empty((V, 6), np.float32); unpool = np.empty((6, V), np.float32)
    m._ctx.call("mocha_graph_constants", *[a.ctypes.data_as(C.c_void_p) for a in (A_j, A_b, pool, unpool)])
    assert np.array_equal(A_j, sd["mot_embedding.2.A_j"]) and np.array_equal(A_b, sd["mot_embedding.5.A_b"])
    assert np.array_equal(pool, sd["mot_embedding.3.weight"]) and np.array_equal(unpool, sd["to_mot.3.weight"])
    p = C.c_void_p()
    m._ctx.call("mocha_pos_emb", C.byref(p))
    host = np.empty((90, 256), np.float32)
    torch.cuda.synchronize()
    hip = C.CDLL("libamdhip64.so.7")                                                          # already loaded by torch: same runtime
    assert hip.hipMemcpy(host.ctypes.data_as(C.c_void_p), p, host.nbytes, 2) == 0          # hipMemcpyDeviceToHost
    assert np.array_equal(host, sd["pos_emb"][0])
    assert torch.equal(m.pos_emb.cpu(), torch.from_numpy(sd["pos_emb"]))


def test_generation_counts_buffer_replacements(model):
    mean, std = _norm()
    ctx = model._ctx
    e, c, n = model.encode(torch.from_numpy(synthetic.pose_windows(1, 6)), mean, std)
    g0 = ctx.generation()
    model.encode(torch.from_numpy(synthetic.pose_windows(2, 6)), mean, std)                  # same size: nothing replaced
    assert ctx.generation() == g0
    bank = ContextBank(model, n, e)                                                          # new current bank
    g1 = ctx.generation()
    assert g1 > g0
    bank.query(n)                                                                            # scratch was sized by bank_set
    bank.characterize(torch.from_numpy(synthetic.pose_windows(3, 6)), mean, std)
    assert ctx.generation() == g1
    model.characterize_pair(torch.from_numpy(synthetic.pose_windows(3, 4)), torch.from_numpy(synthetic.pose_windows(4, 2)), mean, std)
    assert ctx.generation() == g1                                                            # the transient bank does not count
    model.encode(torch.from_numpy(synthetic.pose_windows(2, 40)), mean, std)                 # grows the workspaces
    assert ctx.generation() > g1


def test_streamer_does_not_shrink_the_workspace_and_follows_its_bank(model):
    """the streamer must not lower the workspace limit, and must notice another bank / replaced buffers."""
    mean, std = _norm()
    cha = torch.from_numpy(synthetic.pose_windows(21, 30)).cuda()
    src = torch.from_numpy(synthetic.pose_windows(22, 12)).cuda()
    e, c, n = model.encode(cha, mean, std)
    bank = ContextBank(model, n, e)
    Yb, ib = bank.characterize(src, mean, std, return_index=True)
    sc = StreamingCharacterizer(bank, mean, std)
    y0, i0 = sc.step(src[0]); y0 = y0.clone(); i0 = int(i0.item())
    assert i0 == int(ib[0])
    # a large batch afterwards still runs as one chunk (the limit was not lowered to 1) ...
    Y2 = model.characterize_pair(src, cha, mean, std)
    assert float((Y2 - Yb).abs().max()) < 2e-5
    # ... another bank becomes current, a bigger batch replaces the workspaces: the next step is still this bank's answer
    other = ContextBank(model, n[:3].contiguous(), e[:3].contiguous())
    other.query(n[:2].contiguous())
    model.encode(torch.from_numpy(synthetic.pose_windows(23, 64)), mean, std)
    for i in (0, 5, 11):
        y, idx = sc.step(src[i])
        assert int(idx.item()) == int(ib[i])
        assert float((y - Yb[i]).abs().max()) < 2e-5
    y, _ = sc.step(src[0])
    assert torch.equal(y, y0)                                                                 # same kernels, same window: bit-identical


def test_repeated_load_state_dict_reuses_device_memory():
    sd = weights.synthetic_state_dict(5, 1.0)
    m = Generator(device="cuda:0").load_state_dict(sd).eval()
    X = torch.from_numpy(synthetic.pose_windows(9, 2)).cuda()
    Y0 = m(X, X)
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    for _ in range(5):
        m.load_state_dict(sd)
    torch.cuda.synchronize()
    assert free0 - torch.cuda.mem_get_info()[0] < (4 << 20)                                   # a weight set is ~60 MB with the folded copies
    assert torch.equal(m(X, X), Y0)
    sd2 = weights.synthetic_state_dict(6, 1.0)
    assert not torch.equal(m.load_state_dict(sd2)(X, X), Y0)                                  # and a new set really replaces the old one


def test_non_finite_queries_do_not_fault(model):
    mean, std = _norm()
    e, c, n = model.encode(torch.from_numpy(synthetic.pose_windows(31, 20)), mean, std)
    bank = ContextBank(model, n, e)
    q = n[:12].clone()
    q[3] = float("nan"); q[7] = float("inf")
    d, i = bank.query(q)                                                                      # many-query path (GEMM + arg-min)
    i = i[:, 0].cpu().numpy()
    assert ((i >= 0) & (i < 20)).all()
    ok = [k for k in range(12) if k not in (3, 7)]
    assert i[ok].tolist() == ok
    assert not np.isfinite(d[3, 0].item()) and not np.isfinite(d[7, 0].item())
    g = bank.gather(torch.tensor([-5, 3, 10 ** 6], dtype=torch.int32))                        # out-of-range indices are clamped
    assert torch.equal(g[0], e[0]) and torch.equal(g[1], e[3]) and torch.equal(g[2], e[19])


def test_rccl_bank_broadcast_one_rank(model):
    """mocha_comm_* + mocha_bank_broadcast with a one-rank communicator: RCCL loads, the collective sequence runs, the
    bank is unchanged.  (N > 1 ranks need N GPUs: the driver's scaling run exercises them through bench.py.)"""
    from mocha_sigasia2023_amd import distributed as D
    mean, std = _norm()
    e, c, n = model.encode(torch.from_numpy(synthetic.pose_windows(41, 9)), mean, std)
    bank = ContextBank(model, n, e)
    ref = bank.query(n, return_distance=False)[:, 0].tolist()
    os.environ.pop("RANK", None); os.environ.pop("WORLD_SIZE", None)
    D.init_comm(model)
    got = D.bank_broadcast(model, bank, 9, root=0)
    torch.cuda.synchronize()
    assert got is bank
    assert bank.query(n, return_distance=False)[:, 0].tolist() == ref == list(range(9))


def test_sharded_context_bank_one_rank(model):
    from mocha_sigasia2023_amd.bank import ShardedContextBank
    mean, std = _norm()
    e, c, n = model.encode(torch.from_numpy(synthetic.pose_windows(51, 17)), mean, std)
    sb = ShardedContextBank(model, n.reshape(17, -1), e, 17)
    d, i = sb.query(n[:5].reshape(5, -1))
    assert i.tolist() == [0, 1, 2, 3, 4]
    assert torch.equal(sb.gather(i), e[:5])


@pytest.mark.timeout(900)
def test_bench_one_rank_over_rccl_and_launcher_refuses_missing_gpus():
    """bench.py with the RCCL plumbing forced on (one rank): process group, clip broadcast, C-ABI bank broadcast, max-over-ranks
    timing all run; `--gpus N` beyond the node's GPUs is refused by the launcher before any rank starts."""
    env = dict(os.environ, MOCHA_FORCE_DIST="1")
    env.pop("RANK", None); env.pop("WORLD_SIZE", None)
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--steps", "2", "--warmup", "1", "--no-cpu-baseline",
                        "--no-extras", "--windows", "64"], env=env, capture_output=True, text=True, timeout=800)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])      # RCCL prints its banner to stdout
    assert line["n_gpus"] == 1 and line["bank_broadcast_ms"] is not None and line["bank_broadcast_error"] is None and line["value"] > 0
    assert len(line["per_rank_frames_per_s"]) == 1
    n = torch.cuda.device_count()
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", str(n + 1)], capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "exposes" in (r.stderr + r.stdout)


def test_step_graph_raw_poses(model):
    """mocha_step_graph with raw = 1: un-normalised poses with the root bone in, de-normalised poses out (rows a1 / a13 fused),
    equal to the batched raw call bit for bit (same kernels at B = 1)."""
    from mocha_sigasia2023_amd.generator import _ptr, _stream
    rng = np.random.Generator(np.random.PCG64(8))
    Xm = rng.standard_normal((25, 15)).astype(np.float32); Xs = rng.uniform(0.5, 2.0, (25, 15)).astype(np.float32)
    Ym = rng.standard_normal((25, 15)).astype(np.float32); Ys = rng.uniform(0.5, 2.0, (25, 15)).astype(np.float32)
    model.set_pose_norm(Xm, Xs, Ym, Ys)
    mean, std = _norm()
    tm, ts = torch.from_numpy(mean).cuda(), torch.from_numpy(std).cuda()
    cha = torch.from_numpy((rng.standard_normal((12, 60, 25, 15)) * 2 + 1).astype(np.float32)).cuda()
    src = torch.from_numpy((rng.standard_normal((3, 60, 25, 15)) * 2 + 1).astype(np.float32)).cuda()
    e, c, n = model.encode(cha, mean, std, raw=True)
    bank = ContextBank(model, n, e)
    x = torch.empty((1, 60, 25, 15), device="cuda"); y = torch.empty((1, 60, 24, 15), device="cuda")
    idx = torch.zeros((1,), dtype=torch.int32, device="cuda")
    for i in range(3):
        Yb, ib = bank.characterize(src[i:i + 1].contiguous(), mean, std, return_index=True, raw=True)
        x.copy_(src[i:i + 1])
        model._ctx.call("mocha_step_graph", _ptr(x), _ptr(tm), _ptr(ts), _ptr(y), _ptr(idx), 1, _stream())
        torch.cuda.synchronize()
        assert int(idx.item()) == int(ib.item()) and torch.equal(y, Yb)


@pytest.mark.timeout(900)
def test_bank4k_workload_runs():
    """bench.py --workload bank4k (BASELINE configs[2]): one short run end to end."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--workload", "bank4k", "--steps", "2", "--warmup", "1"],
                       env=env, capture_output=True, text=True, timeout=800)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["scaling"] == "strong" and line["value"] > 1e4 and line["n_gpus"] == 1


def _two_contexts(flag, priority=0, reps=6):
    """Two independent contexts driven from two torch streams at once (their kernels co-reside on the CUs): the embedding, the
    encoder and a whole characterisation of each equal the sequential results bit for bit."""
    sd = weights.synthetic_state_dict(5, 1.3)
    mean, std = _norm()
    bad = []
    m1 = Generator(device="cuda:0").load_state_dict(sd).eval()
    m2 = Generator(device="cuda:0").load_state_dict(sd).eval()
    for m in (m1, m2):
        m.set_option("gemm_bf16x3", flag).set_option("attention_bf16x3", flag)
    X1 = torch.from_numpy(synthetic.pose_windows(3, 75)).to("cuda:0")
    X2 = torch.from_numpy(synthetic.pose_windows(4, 300)).to("cuda:0")
    C1 = torch.from_numpy(synthetic.pose_windows(5, 60)).to("cuda:0")
    C2 = torch.from_numpy(synthetic.pose_windows(6, 200)).to("cuda:0")
    r1, r2 = m1.mot_embedding(X1), m2.mot_embedding(X2)
    e1, e2 = m1.encoder(r1), m2.encoder(r2)
    y1, y2 = m1.characterize_pair(X1, C1, mean, std), m2.characterize_pair(X2, C2, mean, std)
    torch.cuda.synchronize()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream(priority=priority)
    for _ in range(reps):
        with torch.cuda.stream(s1):
            a1 = m1.mot_embedding(X1); o1 = m1.encoder(a1); z1 = m1.characterize_pair(X1, C1, mean, std)
        with torch.cuda.stream(s2):
            a2 = m2.mot_embedding(X2); o2 = m2.encoder(a2); z2 = m2.characterize_pair(X2, C2, mean, std)
        torch.cuda.synchronize()
        errs = tuple(float((x - y).abs().max()) for x, y in ((a1, r1), (a2, r2), (o1, e1), (o2, e2), (z1, y1), (z2, y2)))
        if max(errs) != 0.0:
            bad.append(errs)
    return bad


@pytest.mark.parametrize("planes", [0, 1])
@pytest.mark.parametrize("priority", [0, -1])
def test_two_contexts_on_two_streams(planes, priority):
    """Same-priority streams and streams of different priority (different hardware queues), either engine set.  Round 2 found one
    kernel (mocha_body_front, pointwise.hip) whose results depended on what shared its CU; this test is what reproduces it when it
    runs after the streamer / sharded-bank tests of this file."""
    bad = _two_contexts(planes, priority)
    assert not bad, bad


def test_pair_call_leaves_the_users_bank_image_alone():
    """mocha_characterize_pair's transient bank used to repack the context's packed plane image of the
    centred bank and leave it marked valid, so the next many-query match against the user's fp32 bank ranked the coarse
    scores of the WRONG bank.  Interleaving: set a bank, run a pair call with another bank size, query with Q > 8."""
    from oracle import mocha_oracle as O          # checker only
    model = Generator(device="cuda:0").load_state_dict(weights.synthetic_state_dict(11, 1.0)).eval()
    mean, std = _norm()
    r = np.random.Generator(np.random.PCG64(5))
    bank_nm = r.standard_normal((300, 90, 256)).astype(np.float32)
    q = (bank_nm[r.integers(0, 300, 64)] + 0.05 * r.standard_normal((64, 90, 256))).astype(np.float32)
    bank = ContextBank(model, torch.from_numpy(bank_nm), torch.from_numpy(bank_nm))
    ref_idx, ref_dist = O.match_bruteforce(q, bank_nm)
    d0, i0 = bank.query(torch.from_numpy(q))
    assert np.array_equal(i0[:, 0].cpu().numpy().astype(np.int64), ref_idx)
    model.reserve(128)                                                                        # workspace for the pair below: no growth inside it
    bank.query(torch.from_numpy(q))
    g = model._ctx.generation()
    model.characterize_pair(torch.from_numpy(synthetic.pose_windows(41, 40)), torch.from_numpy(synthetic.pose_windows(42, 77)), mean, std)
    assert model._ctx.generation() == g
    d1, i1 = bank.query(torch.from_numpy(q))                                                  # many-query path on the user's bank again
    assert np.array_equal(i1[:, 0].cpu().numpy().astype(np.int64), ref_idx)
    assert torch.equal(d0, d1)


def test_reloading_weights_moves_the_generation_once_images_exist():
    """finalising weights frees the packed plane images of the GEMM weights; a graph captured after
    warm-up has those pointers baked in, so the generation must move."""
    sd = weights.synthetic_state_dict(5, 1.0)
    m = Generator(device="cuda:0").load_state_dict(sd).eval()
    X = torch.from_numpy(synthetic.pose_windows(9, 64)).cuda()                                # large enough for the plane engine
    m(X, X)
    torch.cuda.synchronize()
    g = m._ctx.generation()
    m.load_state_dict(sd)
    assert m._ctx.generation() > g


def test_build_provenance_is_recorded(model):
    lib = model._ctx.lib
    info = lib.mocha_build_info().decode()
    assert info.startswith("hipcc HIP ") and "gfx950" in info
    assert lib.mocha_runtime_version() > 0


def test_dual_stream_with_lane_sets_never_halves_into_a_small_set():
    """with lanes >= 2 workspace set 1 exists at 8 windows unless the two-stream split needed it whole when the
    sets were planned.  reserve(32) under the default dual_min of 128 (and a dual_min lowered after allocation) used to send the
    second half of a 256-window batch, in 32-window chunks, into that 8-window set: out-of-bounds device writes.  Now the split
    only runs on a full-size set 1, and set_option("dual_min") re-plans the sets."""
    sd = weights.synthetic_state_dict(11, 1.0)
    ref_m = Generator(device="cuda:0").load_state_dict(sd).eval()
    X = torch.from_numpy(synthetic.pose_windows(31, 256)).cuda()
    ref = ref_m(X, X.flip(0))
    m = Generator(device="cuda:0").load_state_dict(sd).eval()
    m.set_option("lanes", 2).set_option("dual_stream", 1)
    m.reserve(32)                                     # chunk 32 < dual_min / 2: set 1 is a lane's 8-window set
    guard = torch.full((1 << 22,), 7.0, device="cuda")               # allocated right after the workspaces: a likely landing place
    Y = m(X, X.flip(0))
    torch.cuda.synchronize()
    assert float((Y - ref).abs().max()) < 1e-5
    assert bool((guard == 7.0).all())
    g = m._ctx.generation()
    m.set_option("dual_min", 16)                      # now the split applies at chunk 32: the sets must be re-planned
    Y2 = m(X, X.flip(0))
    torch.cuda.synchronize()
    assert m._ctx.generation() > g
    assert float((Y2 - ref).abs().max()) < 1e-5
    m.set_option("dual_stream", 0).set_option("lanes", 1)


def test_path_selecting_options_move_the_generation(model):
    """an option that changes which kernels a step launches must invalidate captured step graphs."""
    for name, other in (("scan16", 0), ("fold_joint", 0), ("fold_decoder", 0), ("fold_upsample", 0), ("embed_sums", 0), ("attention_split_max", 0),
                        ("gemm_bf16x3", 0), ("attention_bf16x3", 0)):
        g = model._ctx.generation()
        model.set_option(name, other)
        assert model._ctx.generation() > g, name
    for name, dflt in (("scan16", 1), ("fold_joint", 1), ("fold_decoder", 1), ("fold_upsample", 1), ("embed_sums", 1), ("attention_split_max", 192),
                       ("gemm_bf16x3", 1), ("attention_bf16x3", 1)):
        model.set_option(name, dflt)
