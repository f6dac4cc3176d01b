"""Edge cases of the hot path through the C ABI: empty and single-element batches, a one-entry bank, more queries than
bank rows, a batch that is not a multiple of the internal chunk, the largest bank of BASELINE configs[4] (16 384 rows,
size-independent properties), and misuse that must fail loudly."""
import numpy as np
import pytest
import torch

from mocha_sigasia2023_amd import synthetic, weights

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def model():
    from mocha_sigasia2023_amd import Generator
    return Generator(device="cuda:0").load_state_dict(weights.synthetic_state_dict(11, 1.0)).eval()


def _norm():
    m, s = synthetic.cnt_norm(3)
    return torch.from_numpy(m).cuda(), torch.from_numpy(s).cuda()


def test_empty_batches_return_empty(model):
    from mocha_sigasia2023_amd import ContextBank, mean_variance_norm
    from mocha_sigasia2023_amd.postprocess import PostProcessor, pose_heads
    mean, std = _norm()
    X0 = torch.empty((0, 60, 24, 15), device="cuda")
    T0 = torch.empty((0, 90, 256), device="cuda")
    assert model.mot_embedding(X0).shape == (0, 90, 256)
    assert model.encoder(T0).shape == (0, 90, 256)
    assert model.decoder(T0, T0).shape == (0, 90, 256)
    assert model.to_mot(T0).shape == (0, 60, 24, 15)
    assert model.style_constants(T0).shape == (0, 1024)
    assert model(X0, X0).shape == (0, 60, 24, 15)
    assert mean_variance_norm(T0.permute(0, 2, 1)).shape == (0, 256, 90)
    enc, cnt, nm = model.encode(X0, mean, std)
    assert enc.shape == cnt.shape == nm.shape == (0, 90, 256)
    e1, c1, n1 = model.encode(torch.from_numpy(synthetic.pose_windows(1, 3)), mean, std)
    bank = ContextBank(model, n1, e1)
    Y, idx = bank.characterize(X0, mean, std, return_index=True)
    assert Y.shape == (0, 60, 24, 15) and idx.shape == (0,)
    d, i = bank.query(T0, k=1)
    assert d.shape == (0, 1) and i.shape == (0, 1)
    h, s = pose_heads(model, X0)
    assert h.shape == (0, 24, 13) and s.shape == (0,)
    out = PostProcessor(model).run(np.zeros((0, 24, 13), np.float32), np.zeros(0, np.float32), np.zeros((0, 3), np.float32),
                                   np.zeros((0, 3), np.float32), np.zeros(0, np.float32), np.zeros((0, 2), np.uint8))
    assert out["pos"].shape == (0, 25, 3)


def test_single_window_and_one_entry_bank(model):
    from mocha_sigasia2023_amd import ContextBank
    from oracle import mocha_oracle as O
    mean, std = _norm()
    sd = O.to_torch_state(weights.synthetic_state_dict(11, 1.0))
    src = torch.from_numpy(synthetic.pose_windows(5, 1)); cha = torch.from_numpy(synthetic.pose_windows(6, 1))
    e, c, n = model.encode(cha, mean, std)
    bank = ContextBank(model, n, e)
    Y, idx = bank.characterize(src, mean, std, return_index=True)
    assert idx.tolist() == [0]
    with torch.no_grad():
        Yo = O.generator_forward(sd, src, cha)          # a one-entry bank makes the NN branch equal Generator.forward
    assert float((Y.cpu() - Yo).abs().max()) < 1e-4
    # more queries than bank rows
    q = torch.from_numpy(synthetic.token_features(9, 7)).cuda()
    d, i = bank.query(q, k=1)
    assert i[:, 0].tolist() == [0] * 7
    ref = torch.sqrt(((q.reshape(7, -1) - n.reshape(1, -1)) ** 2).sum(1))
    assert torch.allclose(d[:, 0], ref, rtol=1e-5)


def test_batch_not_a_multiple_of_the_chunk(model):
    """1 031 windows with a 512-window chunk: two full chunks and a 7-window tail that goes through the skinny kernels."""
    mean, std = _norm()
    X = torch.from_numpy(synthetic.pose_windows(8, 1031)).cuda()
    model.reserve(512)
    try:
        enc, cnt, nm = model.encode(X, mean, std)
    finally:
        model.reserve(1280)
    e2, c2, n2 = model.encode(X[1020:].contiguous(), mean, std)
    assert float((enc[1020:] - e2).abs().max()) < 3e-6 * float(e2.abs().max())
    assert torch.isfinite(enc).all()


def test_largest_bank_properties(model):
    """BASELINE configs[4]: a 16 384-entry bank.  Size-independent properties: every bank row finds itself at distance 0,
    the fp32 scan and the many-query GEMM path agree, a perturbed row still finds its origin."""
    from mocha_sigasia2023_amd import ContextBank
    N = 16384
    g = torch.Generator(device="cuda").manual_seed(3)
    nm = torch.randn((N, 90, 256), device="cuda", generator=g)
    enc = torch.empty((1, 90, 256), device="cuda").expand(N, 90, 256)          # never gathered here
    bank = ContextBank(model, nm, nm)
    rows = torch.tensor([0, 1, 4095, 8192, 16383, 777, 12345, 9999, 31, 16000], device="cuda")
    d, i = bank.query(nm[rows].contiguous(), k=1)                               # 10 queries: GEMM path
    assert torch.equal(i[:, 0].long(), rows) and float(d.abs().max()) < 1e-3
    for r in (0, 16383, 5000):                                                  # 1 query: streaming scan
        d1, i1 = bank.query((nm[r:r + 1] + 0.01).contiguous(), k=1)
        assert int(i1[0, 0]) == r
        assert abs(float(d1[0, 0]) - 0.01 * (90 * 256) ** 0.5) < 1e-3
    del enc


def test_misuse_fails_loudly(model):
    from mocha_sigasia2023_amd import ContextBank, Generator
    mean, std = _norm()
    with pytest.raises(TypeError):
        model.encoder(torch.zeros((2, 90, 256), dtype=torch.float64, device="cuda"))
    with pytest.raises(ValueError):
        model.mot_embedding(torch.zeros((2, 60, 22, 15), device="cuda"))      # wrong joint count for this layout
    with pytest.raises(ValueError):
        model.decoder(torch.zeros((2, 90, 256), device="cuda"), torch.zeros((3, 90, 256), device="cuda"))
    fresh = Generator(device="cuda:0")
    with pytest.raises(RuntimeError, match="load_state_dict"):
        fresh.encoder(torch.zeros((1, 90, 256), device="cuda"))
    with pytest.raises(RuntimeError):
        Generator(device="cpu")
    with pytest.raises((RuntimeError, ValueError)):
        ContextBank(model, torch.zeros((0, 90, 256), device="cuda"), torch.zeros((0, 90, 256), device="cuda"))


def test_contexts_release_their_device_memory():
    """Create / use / destroy contexts repeatedly: the library must hand every device allocation back (weights, workspaces,
    bank copies, match scratch, events)."""
    import gc
    from mocha_sigasia2023_amd import ContextBank, Generator
    sd = weights.synthetic_state_dict(2, 1.0)
    mean, std = _norm()
    X = torch.from_numpy(synthetic.pose_windows(1, 24)).cuda()

    def cycle():
        m = Generator(device="cuda:0").load_state_dict(sd).eval()
        e, c, n = m.encode(X, mean, std)
        ContextBank(m, n, e, bf16=True).characterize(X, mean, std)
        ContextBank(m, n, e, copy=True).characterize(X[:3], mean, std)
        m.characterize_pair(X[:10], X[10:], mean, std)
        m.set_option("dual_stream", 1); m(X, X); m.set_option("dual_stream", 0)
        del m
        gc.collect(); torch.cuda.synchronize()

    cycle(); cycle()
    free0 = torch.cuda.mem_get_info()[0]
    for _ in range(5):
        cycle()
    free1 = torch.cuda.mem_get_info()[0]
    assert free0 - free1 < 64 << 20, f"device memory shrank by {(free0 - free1) >> 20} MiB over 5 context life cycles"


def _torch_bruteforce(q, bank):
    """float64 1-NN on the device, ties to the lowest index (reference semantics of the matcher)."""
    q64 = q.double()
    best_d = torch.full((q.shape[0],), float("inf"), dtype=torch.float64, device=q.device)
    best_i = torch.zeros((q.shape[0],), dtype=torch.int64, device=q.device)
    for s in range(0, bank.shape[0], 2048):
        d = torch.cdist(q64, bank[s:s + 2048].double()) ** 2
        dm, im = d.min(dim=1)
        upd = dm < best_d
        best_d = torch.where(upd, dm, best_d); best_i = torch.where(upd, im + s, best_i)
    return best_i, best_d.sqrt()


def test_many_query_matcher_with_more_candidates_than_its_list_holds(model):
    """Every row of the bank ties with every other one (300 copies of one entry, and a second group 1e-3 further away): all of
    them are inside the coarse pass's error bound, more than the select kernel's 128-entry list holds, so it walks the index
    windows and re-evaluates EVERY candidate (passes of 16, the last one partly filled: a row's distance must not depend on how
    many candidates share its pass); ties must go to the lowest index, exactly like the few-query scan and the float64 search."""
    from mocha_sigasia2023_amd import ContextBank
    r = np.random.Generator(np.random.PCG64(3))
    base = r.standard_normal((1, 90 * 256)).astype(np.float32)
    bank = np.repeat(base, 300, axis=0)
    bank[150:] += 1e-3                                              # second group: farther from the queries below
    q = (base + 1e-4 * r.standard_normal((11, 90 * 256))).astype(np.float32)
    tb, tq = torch.from_numpy(bank).cuda(), torch.from_numpy(q).cuda()
    for bf16 in (False, True):
        b = ContextBank(model, tb, tb.view(300, 90, 256), bf16=bf16)
        d, i = b.query(tq)                                          # 11 queries: coarse pass + select
        assert i[:, 0].cpu().tolist() == [0] * 11
        d8, i8 = b.query(tq[:3])                                    # 3 queries: the streaming scan agrees
        assert i8[:, 0].cpu().tolist() == [0] * 3
        assert torch.allclose(d[:3, 0], d8[:, 0], rtol=1e-5)


@pytest.mark.timeout(900)
def test_many_query_matcher_beyond_the_register_resident_score_chunks(model):
    """A bank of more than 16 384 rows (the select kernel keeps four 4 096-row score chunks in registers and recomputes the
    rest), ragged row count, both bank precisions, against a float64 search on the device."""
    from mocha_sigasia2023_amd import ContextBank
    N = 16384 + 517
    g = torch.Generator(device="cuda"); g.manual_seed(99)
    bank = torch.randn((N, 90 * 256), device="cuda", generator=g)
    q = torch.randn((12, 90 * 256), device="cuda", generator=g)
    q[0] = bank[N - 1] + 0.01 * q[0]; q[1] = bank[16384 + 3] + 0.01 * q[1]; q[2] = bank[5] + 0.01 * q[2]
    ri, rd = _torch_bruteforce(q, bank)
    d, i = ContextBank(model, bank, bank.view(N, 90, 256)).query(q)
    assert torch.equal(i[:, 0].long(), ri) and i[0, 0].item() == N - 1 and i[1, 0].item() == 16387
    assert torch.allclose(d[:, 0].double(), rd, rtol=1e-5)
    c = bank.double().mean(0).float()
    b16 = (bank - c).to(torch.bfloat16).float()                      # what the bf16 bank holds: bf16(b - centroid)
    ri16, _ = _torch_bruteforce(q - c, b16)
    i16 = ContextBank(model, bank, bank.view(N, 90, 256), bf16=True).query(q, return_distance=False)
    assert torch.equal(i16[:, 0].long(), ri16)


@pytest.mark.timeout(900)
@pytest.mark.parametrize("Q,N", [(9, 37), (40, 585), (128, 4096), (300, 6200), (700, 1000)])
def test_lazy_selection_is_the_staged_selection(model, Q, N):
    """mocha_match_select2 takes the error bound's row statistics from the producer of the centred queries, reads
    TWO bf16 query planes' scores (bf16 banks), answers a query whose coarse minimum stands alone without touching a bank row when only
    the index is wanted, and otherwise evaluates candidates streamed by the whole workgroup (match_select2.hip), where
    mocha_match_select staged the query row in LDS and worked from one plane.  Same rows scanned, same exact evaluation: indices
    equal (and equal to a float64 search), distances equal to fp32 rounding (the two kernels sum a row's terms in different orders), for
    both bank precisions, with and without distances, ragged N, banks past the 4 096 rows a thread keeps in registers, planted
    near-duplicates (several rows inside even the two-plane bound)."""
    from mocha_sigasia2023_amd import ContextBank
    g = torch.Generator(device="cuda"); g.manual_seed(1000 * Q + N)
    bank = torch.randn((N, 90 * 256), device="cuda", generator=g)
    q = torch.randn((Q, 90 * 256), device="cuda", generator=g)
    for k in range(min(Q, 8)):                                   # near-duplicates: several rows inside the coarse pass's error bound
        row = (k * 7919) % N
        q[k] = bank[row] + 1e-3 * q[k]
        bank[(row + 1) % N] = bank[row] + 2e-3 * torch.randn((90 * 256,), device="cuda", generator=g)
    q[Q - 1] = bank[N - 1]                                       # the last row of the last slice, exactly
    ri, rd = _torch_bruteforce(q, bank)
    for bf16 in (False, True):
        res = {}
        for sel in (1, 0):
            model.set_option("select2", sel)
            model.set_option("match_planes", 2 if (sel and bf16) else 1)      # bf16 banks take the new selection with two query planes only
            cb = ContextBank(model, bank, bank.view(N, 90, 256), bf16=bf16)
            d, i = cb.query(q)
            res[sel] = (d[:, 0].clone(), i[:, 0].clone())
            assert torch.equal(cb.query(q, return_distance=False)[:, 0], i[:, 0])         # index-only calls take the early exit
        model.set_option("select2", 1); model.set_option("match_planes", 1)
        assert torch.equal(res[1][1], res[0][1]), f"bf16={bf16}: {(res[1][1] != res[0][1]).sum().item()} indices differ between the kernels"
        assert torch.allclose(res[1][0], res[0][0], rtol=2e-6, atol=1e-6)
        if not bf16:
            assert torch.equal(res[1][1].long(), ri) and torch.allclose(res[1][0].double(), rd, rtol=1e-5, atol=1e-4)
        assert res[1][1][Q - 1].item() == N - 1 and res[1][0][Q - 1].item() == 0.0 if not bf16 else True


@pytest.mark.timeout(900)
@pytest.mark.parametrize("Q,N", [(9, 300), (128, 4096), (200, 1000), (300, 2500)])
def test_coarse_pass_variants_give_the_same_search(model, Q, N):
    """Round 5: the bf16 bank's coarse pass has three builds - round 4's LDS-DMA kernel (match_nt: non-temporal bank loads, the default for one
    query tile), mocha_match_pass256 on the row-major bank (match_pass = 1) and on its operand-order image (match_pass = 2, variant 16:
    non-temporal), each with one or two query planes.  Different K splits and summation orders of the same products; the selection re-evaluates
    every row inside the coarse bound exactly, so indices are EQUAL and distances agree to fp32 rounding - ragged Q and N, banks smaller and
    larger than a 256-row tile, Q past match_pass_max_q falling back to the round-4 kernel."""
    from mocha_sigasia2023_amd import ContextBank
    g = torch.Generator(device="cuda"); g.manual_seed(31 * Q + N)
    bank = torch.randn((N, 90 * 256), device="cuda", generator=g)
    q = torch.randn((Q, 90 * 256), device="cuda", generator=g)
    for k in range(min(Q, 6)):                                   # near-duplicates inside the one-plane bound
        row = (k * 7919) % N
        q[k] = bank[row] + 1e-3 * q[k]
        bank[(row + 1) % N] = bank[row] + 2e-3 * torch.randn((90 * 256,), device="cuda", generator=g)
    base = dict(match_pass=0, match_nt=1, match_planes=1, match_pass_variant=0)
    ref = None
    try:
        for opts in (dict(), dict(match_nt=0), dict(match_pass=1), dict(match_pass=1, match_pass_variant=3), dict(match_pass=2), dict(match_pass=2, match_pass_variant=16),
                     dict(match_planes=2), dict(match_pass=1, match_planes=2), dict(match_pass=2, match_planes=2, match_pass_variant=16)):
            for k, v in {**base, **opts}.items():
                model.set_option(k, v)
            cb = ContextBank(model, bank, bank.view(N, 90, 256), bf16=True)
            d, i = cb.query(q)
            if ref is None:
                ref = (d[:, 0].clone(), i[:, 0].clone())
            assert torch.equal(i[:, 0], ref[1]), (opts, torch.nonzero(i[:, 0] != ref[1]).flatten().tolist())
            assert torch.allclose(d[:, 0], ref[0], rtol=2e-6, atol=1e-6), opts
    finally:
        for k, v in base.items():
            model.set_option(k, v)


@pytest.mark.timeout(900)
@pytest.mark.parametrize("dup", ["identical", "perturbed"])
def test_selection_with_more_candidates_than_its_list_holds(model, dup):
    """mocha_match_select2 keeps at most 64 candidates per pass; a query with MORE rows inside the coarse pass's error bound
    takes the 'windowed' branch (the list is rebuilt per 64-row window).  Nothing else reaches it: here 150 / 200 bank rows are identical
    to (or within 1e-6 of) the rows two queries sit on - duplicated database clips - among ordinary rows, for an fp32 bank and for a bf16
    bank with two query planes.  The answer must be the LOWEST row among the nearest (float64 search; ties to the lowest index, the
    BallTree / argmin convention of test_fullframework.py:296) and equal to the staged selection's (select2 = 0)."""
    from mocha_sigasia2023_amd import ContextBank
    N, Q = 1500, 12
    g = torch.Generator(device="cuda"); g.manual_seed(77)
    bank = torch.randn((N, 90 * 256), device="cuda", generator=g)
    q = torch.randn((Q, 90 * 256), device="cuda", generator=g)
    crowd_a = torch.arange(40, 40 + 150, device="cuda")                       # 150 consecutive rows: more than two 64-row windows
    crowd_b = torch.randperm(N - 400, device="cuda", generator=g)[:200] + 400    # 200 scattered rows
    base_a, base_b = bank[40].clone(), bank[777].clone()
    if dup == "identical":
        bank[crowd_a] = base_a; bank[crowd_b] = base_b
    else:
        bank[crowd_a] = base_a + 1e-6 * torch.randn((150, 90 * 256), device="cuda", generator=g)
        bank[crowd_b] = base_b + 1e-6 * torch.randn((200, 90 * 256), device="cuda", generator=g)
    q[0] = base_a + 1e-4 * q[0]
    q[1] = base_b + 1e-4 * q[1]
    q[2] = base_a                                                              # exactly on the crowd
    for bf16 in (False, True):
        # what the search runs over: the fp32 rows, or the bf16-rounded centred rows (the bank the bf16 matcher is exact over)
        res = {}
        for sel in (1, 0):
            model.set_option("select2", sel)
            model.set_option("match_planes", 2 if (sel and bf16) else 1)
            cb = ContextBank(model, bank, bank.view(N, 90, 256), bf16=bf16)
            d, i = cb.query(q)
            res[sel] = (d[:, 0].clone(), i[:, 0].clone())
            assert torch.equal(cb.query(q, return_distance=False)[:, 0], i[:, 0])
        model.set_option("select2", 1); model.set_option("match_planes", 1)
        assert torch.allclose(res[1][0], res[0][0], rtol=2e-6, atol=1e-6)
        crowd = {0: set(crowd_a.tolist()), 1: set(crowd_b.tolist()), 2: set(crowd_a.tolist())}
        for k, rows in crowd.items():
            assert int(res[1][1][k]) in rows and int(res[0][1][k]) in rows, (k, int(res[1][1][k]), int(res[0][1][k]))
        assert torch.equal(res[1][1][3:], res[0][1][3:])                              # the ordinary queries: one answer
        if dup == "identical":
            # exact ties: both kernels must return the LOWEST of the identical rows, for either bank precision
            assert torch.equal(res[1][1], res[0][1]), f"bf16={bf16}: select2 and the staged selection disagree at {torch.nonzero(res[1][1] != res[0][1]).flatten().tolist()}"
            assert int(res[1][1][0]) == 40 and int(res[1][1][2]) == 40 and int(res[1][1][1]) == int(crowd_b.min())
        if not bf16:
            # float64 search in the direct form (no cancellation), ties to the lowest row
            got = res[1][1].long()
            for k in range(Q):
                d64 = ((q[k].double()[None] - bank.double()) ** 2).sum(1).sqrt()
                best = d64.min()
                # rows 1e-6 apart cannot be separated beyond fp32 rounding of the direct-form sum: the winner must be AS NEAR as the float64
                # winner to fp32 accuracy; where rows are identical it is the lowest of them (checked above)
                assert d64[got[k]] <= best * (1 + 1e-5) + 1e-6, (k, int(got[k]), float(d64[got[k]]), float(best))
                if dup == "identical":
                    assert int(got[k]) == int(torch.nonzero(d64 == best).flatten()[0])


@pytest.mark.parametrize("bf16", [False, True])
def test_top_k_query_and_soft_blend(model, bf16):
    """BallTree.query(k > 1) semantics (optional): the k nearest rows, exact, distances ascending, ties to the
    lower index; fewer rows than k -> -1 / inf; and the softmax-weighted blend of the neighbours' encoded entries."""
    from mocha_sigasia2023_amd import ContextBank
    N, Q, k = 203, 11, 5
    g = torch.Generator(device="cuda"); g.manual_seed(5)
    bank = torch.randn((N, 90 * 256), device="cuda", generator=g)
    enc = torch.randn((N, 90, 256), device="cuda", generator=g)
    bank[17] = bank[3]                                             # an exact tie: rows 3 and 17
    q = torch.randn((Q, 90 * 256), device="cuda", generator=g)
    q[0] = bank[3] + 0.01 * q[0]
    b = ContextBank(model, bank, enc, bf16=bf16)
    dist, idx = b.query(q, k=k)
    if bf16:
        c = bank.double().mean(0).float()
        ref_bank, ref_q = (bank - c).to(torch.bfloat16).double(), (q - c).double()
    else:
        ref_bank, ref_q = bank.double(), q.double()
    d = torch.cdist(ref_q, ref_bank)
    order = torch.argsort(d + 1e-15 * torch.arange(N, device="cuda")[None], dim=1, stable=True)[:, :k]
    assert torch.equal(idx.long(), order)
    assert idx[0, 0].item() == 3 and idx[0, 1].item() == 17         # the tie goes to the lower index first
    assert torch.allclose(dist.double(), torch.gather(d, 1, order), rtol=1e-5)
    d1, i1 = b.query(q)                                             # k = 1 fast path agrees with the first column
    assert torch.equal(i1[:, 0], idx[:, 0]) and torch.allclose(d1[:, 0], dist[:, 0], rtol=1e-5)
    out = b.gather_blend(idx, dist, temperature=2.0)
    w = torch.softmax(-dist.double() / 2.0, dim=1)
    ref = (w[:, :, None, None] * enc.double()[idx.long()]).sum(1)
    assert float((out.double() - ref).abs().max()) < 1e-5
    small = ContextBank(model, bank[:3].contiguous(), enc[:3].contiguous(), bf16=bf16)
    ds, js = small.query(q[:2], k=5)                                # fewer rows than k
    assert (js[:, 3:] == -1).all() and torch.isinf(ds[:, 3:]).all() and (js[:, :3] >= 0).all()
    assert torch.isfinite(small.gather_blend(js, ds)).all()
    with pytest.raises(ValueError):
        b.query(q, k=0)


@pytest.mark.timeout(900)
def test_few_query_scan_through_the_bf16_copy_is_the_fp32_search(model):
    """fp32 banks of >= 4096 rows are scanned through their centred bf16 copy (half the HBM bytes) and the rows the rounding bound
    cannot exclude are re-evaluated exactly on the fp32 rows (option "scan16", default on).  Same indices and distances as the
    scan of the fp32 rows themselves and as a float64 search, on: a random bank; rows clustered far from the origin with
    near-duplicates (many candidates); groups of identical rows (every member a candidate: ties to the lowest index, index
    windows); a non-finite query."""
    from mocha_sigasia2023_amd import ContextBank
    r = np.random.Generator(np.random.PCG64(41))
    D, N = 90 * 256, 5000
    rnd = r.standard_normal((N, D)).astype(np.float32)
    off = (3.0 * r.standard_normal((1, D))).astype(np.float32)
    near = (off + 0.02 * r.standard_normal((N, D))).astype(np.float32)
    groups = np.repeat((off + 0.02 * r.standard_normal((50, D))).astype(np.float32), 100, axis=0)      # 50 groups of 100 identical rows
    for name, bank in (("random", rnd), ("near-duplicates", near), ("identical groups", groups)):
        pick = r.integers(0, N, 8)
        q = (bank[pick] + (0.05 if name == "random" else 0.004) * r.standard_normal((8, D))).astype(np.float32)
        tb, tq = torch.from_numpy(bank).cuda(), torch.from_numpy(q).cuda()
        ri, rd = _torch_bruteforce(tq, tb)
        model.set_option("scan16", 1)
        d1, i1 = ContextBank(model, tb, tb.view(N, 90, 256)).query(tq)
        model.set_option("scan16", 0)
        d0, i0 = ContextBank(model, tb, tb.view(N, 90, 256)).query(tq)
        model.set_option("scan16", 1)
        i1, i0 = i1[:, 0].cpu().numpy().astype(np.int64), i0[:, 0].cpu().numpy().astype(np.int64)
        ri = ri.cpu().numpy()
        d64 = torch.cdist(tq.double(), tb.double())
        for k in range(8):                                      # equal, or a float64 near-tie that fp32 cannot separate
            assert i1[k] == ri[k] or abs(float(d64[k, i1[k]] - d64[k, ri[k]])) <= 2e-6 * float(d64[k, ri[k]]), (name, k, i1[k], ri[k])
            assert i0[k] == ri[k] or abs(float(d64[k, i0[k]] - d64[k, ri[k]])) <= 2e-6 * float(d64[k, ri[k]]), (name, k, i0[k], ri[k])
        if name == "identical groups":
            assert (i1 % 100 == 0).all() and np.array_equal(i1, i0)                 # the first member of the nearest group
        assert torch.allclose(d1, d0, rtol=1e-5) and torch.allclose(d1[:, 0].double(), rd, rtol=1e-5)
        for Q in (1, 2, 3, 8):                                   # every instance of the scan kernel
            dq, iq = ContextBank(model, tb, tb.view(N, 90, 256)).query(tq[:Q])
            assert np.array_equal(iq[:, 0].cpu().numpy().astype(np.int64), i1[:Q])
        # the re-rank runs as 16 workgroups per query, each on a slice of the rows; the last to arrive (a ticket that only counts up,
        # kept in the bank's scratch) picks the winner: many calls on one bank, with varying query counts in between, same answers
        bank_obj = ContextBank(model, tb, tb.view(N, 90, 256))
        for rep in range(35):
            dq, iq = bank_obj.query(tq[: 1 + rep % 8])
            assert np.array_equal(iq[:, 0].cpu().numpy().astype(np.int64), i1[: 1 + rep % 8]), (name, rep)
    bad = torch.from_numpy(rnd[:3].copy()).cuda(); bad[1, 7] = float("nan")
    d, i = ContextBank(model, torch.from_numpy(rnd).cuda(), torch.from_numpy(rnd).cuda().view(N, 90, 256)).query(bad)
    i = i[:, 0].cpu().tolist()
    assert i[0] == 0 and i[2] == 2 and 0 <= i[1] < N and not np.isfinite(d[1, 0].item())


def test_null_pointers_come_back_as_error_codes(model):
    """A NULL where a device pointer is required is an argument error of the C ABI (status + message), not a GPU fault that takes
    the host process down; with an empty batch the pointers may be NULL.  Called below the Python mirror, straight at the ABI."""
    from mocha_sigasia2023_amd.generator import _ptr, _stream
    x = torch.zeros((2, 60, model.V, 15), device="cuda:0")
    tok = torch.zeros((2, 90, 256), device="cuda:0")
    ctx = model._ctx
    cases = [("mocha_embed", (_ptr(None), 2, _ptr(tok), 1, _stream())), ("mocha_embed", (_ptr(x), 2, _ptr(None), 1, _stream())),
             ("mocha_encoder", (_ptr(tok), 2, _ptr(None), _stream())), ("mocha_decoder", (_ptr(tok), _ptr(None), 2, _ptr(tok), _stream())),
             ("mocha_to_mot", (_ptr(None), 2, _ptr(x), _stream())), ("mocha_style_constants", (_ptr(tok), 2, _ptr(None), _stream())),
             ("mocha_style_constants", (_ptr(None), 2, _ptr(tok), _stream())), ("mocha_forward", (_ptr(x), _ptr(x), 2, _ptr(None), _stream())),
             ("mocha_encode", (_ptr(x), 2, _ptr(None), _ptr(None), _ptr(None), _ptr(None), _ptr(None), _stream()))]
    for name, args in cases:
        with pytest.raises(RuntimeError, match="null argument"):
            ctx.call(name, *args)
    ctx.call("mocha_encoder", _ptr(None), 0, _ptr(None), _stream())              # empty batch: nothing is touched
    ctx.call("mocha_forward", _ptr(None), _ptr(None), 0, _ptr(None), _stream())
    # cnt is optional in mocha_encode: the z-scored copy alone (a bank build needs nothing else); it equals the one written beside cnt
    X = torch.from_numpy(synthetic.pose_windows(3, 4, model.V)).cuda()
    mean, std = (torch.from_numpy(a).cuda() for a in synthetic.cnt_norm(3))
    enc, nm = torch.empty((4, 90, 256), device="cuda:0"), torch.empty((4, 90, 256), device="cuda:0")
    ctx.call("mocha_encode", _ptr(X), 4, _ptr(enc), _ptr(None), _ptr(mean), _ptr(std), _ptr(nm), _stream())
    e2, _, nm2 = model.encode(X, mean, std)
    assert torch.equal(enc, e2) and torch.equal(nm, nm2)
    with pytest.raises(RuntimeError, match="needs cnt_mean"):
        ctx.call("mocha_encode", _ptr(X), 4, _ptr(enc), _ptr(None), _ptr(None), _ptr(None), _ptr(nm), _stream())


@pytest.mark.parametrize("N", [4096, 4097, 20011, 40000, 70001])
def test_bf16_copy_scan_at_odd_bank_sizes(model, N):
    """The re-rank's branches by bank size: a slice's keys held in registers (up to 2 048 rows per slice: N <= 32 768) or re-read
    (beyond), one index window per slice or several (more than 4 096 rows per slice: N > 65 536), row counts that are not a
    multiple of the scan's 16 rows per workgroup or of the 16 slices, the winner in the first / last row and in the last slice's
    ragged end, and near-duplicates of the winner spread over the slices.  Against the fp32 scan and a float64 search."""
    from mocha_sigasia2023_amd import ContextBank
    D = 90 * 256
    g = torch.Generator(device="cuda:0"); g.manual_seed(N)
    bank = torch.randn((N, D), device="cuda:0", generator=g)
    pick = torch.tensor([0, N - 1, N // 2, N - 3, 17, (N * 15) // 16 + 1, N // 16, 4095], device="cuda:0")
    q = bank[pick] + 0.05 * torch.randn((8, D), device="cuda:0", generator=g)
    # near-duplicates of query 2's winner in other slices: farther by 1e-3 of the distance - candidates the bound cannot exclude
    for j in range(1, 16, 3):
        r = (N // 2 + j * (N // 16)) % N
        if r not in pick.tolist():
            bank[r] = q[2] + 1.001 * (bank[N // 2] - q[2])
    ri, rd = _torch_bruteforce(q, bank)
    assert torch.equal(ri, pick.long())
    model.set_option("scan16", 1)
    b1 = ContextBank(model, bank, bank.view(N, 90, 256))
    d1, i1 = b1.query(q)
    for Q in (1, 3):
        dq, iq = b1.query(q[:Q])
        assert torch.equal(iq, i1[:Q]) and torch.equal(dq, d1[:Q])
    del b1
    model.set_option("scan16", 0)
    d0, i0 = ContextBank(model, bank, bank.view(N, 90, 256)).query(q)
    model.set_option("scan16", 1)
    assert torch.equal(i1[:, 0].long(), ri) and torch.equal(i0[:, 0].long(), ri)
    assert torch.allclose(d1, d0, rtol=1e-5) and torch.allclose(d1[:, 0].double(), rd, rtol=1e-5)


def test_coarse_pass_with_folded_slabs_names_the_same_rows():
    """Option "match_fold" (round 6, a measured negative kept reproducible: DESIGN.md section 8.3): the bf16 coarse pass's last-arriving K-slab
    workgroup per tile leaves the slabs' sum in slab 0 and the selection reads one slab - the sum is added in the selection's own order, so
    indices and distances are those of the unfolded call, bit for bit; the tickets return to zero (a second call agrees too)."""
    import torch
    from mocha_sigasia2023_amd import ContextBank, Generator, weights
    dev = torch.device("cuda:0")
    model = Generator(device=dev).load_state_dict(weights.synthetic_state_dict(3, 1.0)).eval()
    g = torch.Generator(device=dev); g.manual_seed(77)
    N, Dm = 1000, 90 * 256
    bank = torch.randn((N, Dm), device=dev, generator=g)
    out = {}
    for Q in (40, 200):
        q = bank[torch.randperm(N, device=dev, generator=g)[:Q]] + 0.3 * torch.randn((Q, Dm), device=dev, generator=g)
        for fold in (0, 1):
            model.set_option("match_fold", fold)
            b = ContextBank(model, bank, bank.view(N, 90, 256), bf16=True)
            r = [b.query(q) for _ in range(2)]
            assert torch.equal(r[0][1], r[1][1]) and torch.equal(r[0][0], r[1][0])
            out[(Q, fold)] = r[0]
        model.set_option("match_fold", 0)
        assert torch.equal(out[(Q, 0)][1], out[(Q, 1)][1]) and torch.equal(out[(Q, 0)][0], out[(Q, 1)][0])
