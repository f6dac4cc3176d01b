"""The few-query matcher's one-byte first stage (option "scan8", match_scan8.hip), run with -m gpu.

An fp32 bank's centred rows as biased bytes with a per-row scale and a measured residual bound; the scan reads 1 B per value, the refine
re-evaluates on the fp32 rows every row the byte image cannot exclude: the RESULT is the exact fp32 search's - indices and distances equal,
bit for bit, to the round-3 path through the bf16 copy (same refine kernel, same exact evaluation) and to a float64 search.  The stage
is adaptive on the device: on a bank whose rows the image cannot separate it switches itself off for the following calls.
Matching semantics: BallTree(k=1), test_fullframework.py:293-298, 440-443."""
import ctypes as C

import numpy as np
import pytest
import torch

from mocha_sigasia2023_amd import ContextBank, Generator, StreamingCharacterizer, synthetic, weights

pytestmark = pytest.mark.gpu
D = 90 * 256


def dev():
    return torch.device("cuda:0")


def scan8_state(model):
    st = (C.c_int32 * 2)()
    model._ctx.call("mocha_scan_byte_state", 0, st, None)
    return st[0], st[1]


@pytest.fixture(scope="module")
def model():
    return Generator(layout="mixamo", device=dev()).load_state_dict(weights.synthetic_state_dict(1777, 1.0, "mixamo")).eval()


def _f64_search(q, bank):
    d2 = torch.cdist(q.double(), bank.double())
    return d2.argmin(1).cpu().numpy(), d2.min(1).values.cpu().numpy()


@pytest.mark.parametrize("Q", [1, 2, 3, 4])
def test_planted_queries_keep_the_byte_stage_and_the_exact_answer(model, Q):
    """Queries a small step away from rows of the bank (gap to every other row ~ 215, the byte image's bound ~ 3): a handful of
    candidates, the stage stays on, answers == the bf16-copy path == float64."""
    g = torch.Generator(device=dev()); g.manual_seed(80 + Q)
    N = 6000
    bank = 1.5 * torch.randn((N, D), device=dev(), generator=g) + 0.3
    rows = torch.randperm(N, device=dev(), generator=g)[:Q]
    q = bank[rows] + 0.02 * torch.randn((Q, D), device=dev(), generator=g)
    out = {}
    for on in (0, 1):
        model.set_option("scan8", on)
        b = ContextBank(model, bank, bank.view(N, 90, 256))
        assert scan8_state(model)[0] == on
        for rep in range(3):
            dist, idx = b.query(q)
        out[on] = (idx.clone(), dist.clone())
        if on:
            assert scan8_state(model) == (1, 0), "planted queries must not switch the byte stage off"
    model.set_option("scan8", 0)
    assert torch.equal(out[0][0], out[1][0]) and torch.equal(out[0][1], out[1][1])
    ri, rd = _f64_search(q, bank)
    assert np.array_equal(out[1][0][:, 0].cpu().numpy(), ri) and np.array_equal(ri, rows.cpu().numpy())
    assert np.allclose(out[1][1][:, 0].cpu().numpy(), rd, rtol=1e-5)


def test_random_bank_switches_the_stage_off_and_stays_exact(model):
    """Independent N(0, 1) rows: every distance is 214.7 +- 1, thousands of rows lie inside the byte image's bound.  The first call is
    still exact (every candidate is re-evaluated), sets the sticky mode word, and the following calls run the bf16 scan; a new bank
    clears the word."""
    g = torch.Generator(device=dev()); g.manual_seed(90)
    N = 8192
    bank = torch.randn((N, D), device=dev(), generator=g)
    q = torch.randn((2, D), device=dev(), generator=g)
    model.set_option("scan8", 0)
    b0 = ContextBank(model, bank, bank.view(N, 90, 256))
    d0, i0 = b0.query(q)
    model.set_option("scan8", 1)
    b1 = ContextBank(model, bank, bank.view(N, 90, 256))
    assert scan8_state(model)[0] == 1
    d1, i1 = b1.query(q)                                   # byte scan + a long refine
    assert scan8_state(model) == (1, 1), "thousands of candidates: the stage must have switched itself off"
    d2, i2 = b1.query(q)                                   # bf16 scan from here on
    assert scan8_state(model) == (1, 1)
    for d, i in ((d1, i1), (d2, i2)):
        assert torch.equal(i, i0) and torch.equal(d, d0)
    ri, _ = _f64_search(q, bank)
    assert np.array_equal(i0[:, 0].cpu().numpy(), ri)
    planted = bank.clone(); planted[17] = q[0] + 0.01
    b2 = ContextBank(model, planted, planted.view(N, 90, 256))      # mocha_bank_set: another chance
    assert scan8_state(model) == (1, 0)
    d3, i3 = b2.query(q[:1])
    assert int(i3[0, 0]) == 17 and scan8_state(model) == (1, 0)
    model.set_option("scan8", 0)


def test_streamed_step_graph_with_the_byte_stage(model):
    """BASELINE configs[4] in small: the captured per-window step (mocha_step_graph) with the byte stage inside replays to the indices and
    poses of the stage-less graph, window by window, on a planted bank."""
    NB, W = 4096, 24
    mean, std = synthetic.cnt_norm(7)
    g = torch.Generator(device=dev()); g.manual_seed(7)
    bank_nm = torch.randn((NB, D), device=dev(), generator=g)
    src = torch.from_numpy(synthetic.pose_windows(5, W, 22)).to(dev())
    nm0 = model.encode(src, mean, std)[2].reshape(W, D)
    rows = torch.randperm(NB, device=dev(), generator=g)[:W]
    bank_nm[rows] = nm0 + 0.01 * torch.randn((W, D), device=dev(), generator=g)
    res = {}
    for on in (0, 1):
        model.set_option("scan8", on)
        bank = ContextBank(model, bank_nm, bank_nm.view(NB, 90, 256))
        sc = StreamingCharacterizer(bank, mean, std, use_graph=True)
        ys, ix = [], []
        for i in range(W):
            y, idx = sc.step(src[i])
            ys.append(y.clone()); ix.append(int(idx.item()))
        res[on] = (torch.stack(ys), ix)
        if on:
            assert scan8_state(model) == (1, 0)
    model.set_option("scan8", 0)
    assert res[1][1] == res[0][1] == rows.cpu().tolist()
    assert torch.equal(res[1][0], res[0][0])


def test_byte_stage_edge_cases_ties_outliers_zero_rows_ragged_size(model):
    """Ragged bank size (not a multiple of the scan's 16 rows per workgroup, nor of the refine's 64 slices), identical rows (ties go to the
    LOWEST index, as in the bf16-copy path and BallTree's first hit), an all-zero row (scale 1), a row with a 1e4 outlier (its scale swallows
    every other element: a huge residual bound, the row is always re-ranked): answers equal to the stage-less path bit for bit and to the
    planted rows.  (A NaN row is not a case: it poisons the bank's centroid in either path, as it would BallTree's construction.)"""
    g = torch.Generator(device=dev()); g.manual_seed(123)
    N = 4096 + 37
    bank = torch.randn((N, D), device=dev(), generator=g)
    bank[100] = 0.0
    bank[200, 5] = 1e4
    bank[1500] = bank[1200]                                   # a tie: the answer must be 1200
    bank[N - 1] = bank[7] + 0.5                               # the last (ragged) row is somebody's neighbour
    q = torch.stack([bank[1200] + 0.01 * torch.randn(D, device=dev(), generator=g),
                     torch.zeros(D, device=dev()) + 1e-3,
                     bank[200] + 0.01,
                     bank[N - 1] + 0.001])
    out = {}
    for on in (0, 1):
        model.set_option("scan8", on)
        b = ContextBank(model, bank, bank.view(N, 90, 256))
        out[on] = b.query(q)
    model.set_option("scan8", 0)
    assert torch.equal(out[0][1], out[1][1]) and torch.equal(out[0][0], out[1][0])
    assert out[1][1][:, 0].tolist() == [1200, 100, 200, N - 1]
