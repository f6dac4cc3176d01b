"""Full-size GPU parity (run with -m gpu): every single-GPU BASELINE.json configuration at its real size, through the C ABI,
against the CPU oracle.

* configs[1]  the demo pair, 585 source + 585 character windows (the workload bench.py times, same seeds), 22 and 24 joints,
              through mocha_characterize_pair and through the three-call path, on the DEFAULT engines: every instance of
              the plane GEMM mocha_gemm_x3 (plain / gathered, residual / GELU / LeakyReLU epilogues, the matcher's K-split
              coarse pass), both mocha_attention_x3 instances and every pointwise kernel at the batch size the bench runs
              them at (the exact-f32 engines are compared with these in tests/test_gemm_engines.py).  Reference call sites:
              test_fullframework.py:188-194, 271-277, 293-298, 438-443, 465-467.
* configs[2]  1024 source windows against a 4096-entry bf16 bank: indices against a float64 brute-force search over the
              bf16-rounded centred bank (the bank the kernel actually scans), agreement rate with the fp32 search reported.
* configs[4]  a 300-frame clip streamed window by window (285 windows) against a 16 384-entry bank through the captured
              per-window step (mocha_step_graph): graph replay == batched call bit for bit, indices against the oracle.

Tolerance (north star): |Y - Y_oracle| < 1e-4 ABSOLUTE per joint and channel; nearest-neighbour indices equal, where a
differing index is accepted only if it is a tie at fp32 feature precision judged in float64 on the oracle's own features."""
import numpy as np
import pytest
import torch

from mocha_sigasia2023_amd import ContextBank, Generator, StreamingCharacterizer, synthetic, weights
from oracle import mocha_oracle as O

pytestmark = pytest.mark.gpu
TOL = 1e-4
D = 90 * 256


def dev():
    assert torch.cuda.is_available(), "GPU tests need a ROCm device"
    return torch.device("cuda:0")


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev())


def _bf16_round(a):
    u = np.ascontiguousarray(a, dtype=np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) >> 16
    return (u.astype(np.uint32) << 16).view(np.float32)


def _ties_ok(ours, best, q64, k64, rtol):
    """Every index that differs from the oracle's must be as near as the oracle's winner up to `rtol` (float64, oracle features)."""
    diff = np.nonzero(ours != best)[0]
    for i in diff:
        d_o = np.sqrt(((q64[i] - k64[ours[i]]) ** 2).sum())
        d_b = np.sqrt(((q64[i] - k64[best[i]]) ** 2).sum())
        if not d_o <= d_b * (1 + rtol):
            return False, int(i)
    return True, len(diff)


@pytest.mark.timeout(1800)
@pytest.mark.parametrize("V,layout", [(22, "mixamo"), (24, "mocha")])
def test_config1_demo_pair_full_size(V, layout):
    W = 585
    sd = weights.synthetic_state_dict(1777, 1.0, layout)                    # bench.py's model, inputs and cnt norm
    model = Generator(layout=layout, device=dev()).load_state_dict(sd).eval()
    src = synthetic.pose_windows(1777, W, V)
    cha = synthetic.pose_windows(4242, W, V)
    mean, std = synthetic.cnt_norm(7)
    Yp, ip, enc_c, nm_c = model.characterize_pair(T(src), T(cha), mean, std, return_index=True, return_bank=True)
    e3, _, n3 = model.encode(T(cha), mean, std)
    Y3, i3 = ContextBank(model, n3, e3).characterize(T(src), mean, std, return_index=True)
    torch.cuda.synchronize()
    ost = O.to_torch_state(sd)
    with torch.no_grad():                                                       # O.characterize, keeping its features for the tie check
        def enc_all(X):
            e, c = zip(*(O.encode(ost, torch.from_numpy(X[s:s + 32])) for s in range(0, W, 32)))
            return torch.cat(e), torch.cat(c)
        se, sc_ = enc_all(src)
        ce, cc = enc_all(cha)
        q64 = O.znorm(sc_.numpy(), mean, std).reshape(W, -1).astype(np.float64)
        k64 = O.znorm(cc.numpy(), mean, std).reshape(W, -1).astype(np.float64)
        io, _ = O.match_bruteforce(q64, k64)
        sel = ce[torch.from_numpy(io)]
        Yo = torch.cat([O.to_mot(ost, O.decoder(ost, se[s:s + 32], sel[s:s + 32])) for s in range(0, W, 32)])
        enc_o = ce.numpy()
    Yo = Yo.numpy()
    for name, Y, idx in (("pair", Yp, ip), ("three-call", Y3, i3)):
        ours = idx.cpu().numpy().astype(np.int64)
        ok, info = _ties_ok(ours, io, q64, k64, 1e-6)
        assert ok, f"{name}: query {info} matched a row that is not a nearest neighbour"
        same = ours == io
        assert same.mean() > 0.99, f"{name}: only {same.sum()} of {W} indices equal the oracle's"
        err = np.abs(Y.cpu().numpy()[same] - Yo[same]).max()
        assert err < TOL, f"{name}: max |Y - oracle| = {err:.3e} (absolute)"
        assert torch.isfinite(Y).all()
    # the bank the pair call returns is the character clip's encoding
    assert np.abs(enc_c.cpu().numpy() - enc_o).max() < TOL * max(1.0, np.abs(enc_o).max())
    assert np.abs(nm_c.cpu().numpy().reshape(W, -1) - k64).max() < 6 * TOL * max(1.0, np.abs(k64).max())   # / (std / weight): x up to 6


@pytest.mark.timeout(1800)
def test_config2_1024_windows_x_4k_bf16_bank():
    W, NB, V = 1024, 4096, 22
    sd = weights.synthetic_state_dict(1777, 1.0, "mixamo")
    model = Generator(layout="mixamo", device=dev()).load_state_dict(sd).eval()
    src = T(synthetic.pose_windows(1, W, V))
    mean, std = synthetic.cnt_norm(7)
    g = torch.Generator(device=dev()); g.manual_seed(2)                       # bench.py --workload bank4k
    bank_nm = torch.randn((NB, D), device=dev(), generator=g)
    bank_enc = torch.randn((NB, 90, 256), device=dev(), generator=g)
    b16 = ContextBank(model, bank_nm, bank_enc, bf16=True)
    Y16, i16 = b16.characterize(src, mean, std, return_index=True)
    enc_s, cnt_s, nm_s = model.encode(src, mean, std)                          # the queries the matcher saw (HIP features)
    d16, j16 = b16.query(nm_s)
    assert torch.equal(j16[:, 0], i16)                                         # characterize == encode + query
    b32 = ContextBank(model, bank_nm, bank_enc)
    i32 = b32.query(nm_s, return_distance=False)[:, 0].cpu().numpy()
    torch.cuda.synchronize()
    # float64 brute force over the bank the kernel scans: bf16(bank - centroid) against the centred fp32 queries
    bank = bank_nm.cpu().numpy()
    c = bank.astype(np.float64).mean(0).astype(np.float32)
    q = nm_s.cpu().numpy().reshape(W, -1)
    ridx, rdist = O.match_bruteforce(q - c, _bf16_round(bank - c))
    ours = i16.cpu().numpy().astype(np.int64)
    assert np.array_equal(ours, ridx), f"{(ours != ridx).sum()} of {W} bf16-bank indices differ from the float64 search"
    assert np.allclose(d16[:, 0].cpu().numpy(), rdist, rtol=1e-5)
    ridx32, _ = O.match_bruteforce(q, bank)
    assert np.array_equal(i32, ridx32)                                          # fp32 bank: exact
    agree = float((ours == ridx32).mean())
    print(f"configs[2]: bf16-bank search agrees with the fp32 search on {agree * 100:.2f} % of {W} queries")
    assert agree >= 0.95
    # decode parity on a sample of the windows against the oracle fed with the same matches
    ost = O.to_torch_state(sd)
    sel = np.arange(0, W, 16)
    with torch.no_grad():
        eo, _ = O.encode(ost, src.cpu()[sel])
        Yo = O.to_mot(ost, O.decoder(ost, eo, bank_enc.cpu()[torch.from_numpy(ours[sel])])).numpy()
    assert np.abs(Y16.cpu().numpy()[sel] - Yo).max() < TOL


@pytest.mark.timeout(1800)
def test_config4_streamed_clip_x_16k_bank():
    NB, V, W = 16384, 22, 285
    sd = weights.synthetic_state_dict(1777, 1.0, "mixamo")
    model = Generator(layout="mixamo", device=dev()).load_state_dict(sd).eval()
    mean, std = synthetic.cnt_norm(7)
    g = torch.Generator(device=dev()); g.manual_seed(7)
    bank_nm = torch.randn((NB, D), device=dev(), generator=g)
    src = T(synthetic.pose_windows(5, W, V))
    ost = O.to_torch_state(sd)
    # against pure noise every window would match the same row: plant noisy copies of the windows' own features at scattered rows, so
    # that the 285 answers are 285 different rows (and near-ties exist: two copies per window, the farther one 1e-3 further away)
    _, _, nm0 = model.encode(src, mean, std)
    rows = torch.randperm(NB, device=dev(), generator=g)[: 2 * W]
    nm0 = nm0.reshape(W, D)
    gap = (torch.cdist(nm0, nm0) + 1e30 * torch.eye(W, device=dev())).min().item()         # the closest two windows of the clip
    noise = (0.1 * gap / D ** 0.5) * torch.randn((W, D), device=dev(), generator=g)     # |noise| = a tenth of that
    bank_nm[rows[:W]] = nm0.reshape(W, D) + noise
    bank_nm[rows[W:]] = nm0.reshape(W, D) + noise * 1.001
    planted = rows[:W].cpu().numpy()
    print(f"config4: closest two windows {gap:.3f} apart, feature norm {nm0.norm(dim=1).mean().item():.1f}")
    for bf16 in (False, True):
        bank = ContextBank(model, bank_nm, bank_nm.view(NB, 90, 256), bf16=bf16)
        Yb, ib = bank.characterize(src, mean, std, return_index=True)          # batched: the many-query (GEMM) matcher
        sc = StreamingCharacterizer(bank, mean, std, use_graph=True)
        eager = StreamingCharacterizer(bank, mean, std, use_graph=False)
        idx_stream = np.empty(W, np.int64)
        Ys = torch.empty_like(Yb)
        for i in range(W):
            y, idx = sc.step(src[i])
            Ys[i] = y; idx_stream[i] = int(idx.item())
            if i % 40 == 0:                                                    # replay == eager launch sequence, bit for bit
                y2, idx2 = eager.step(src[i])
                assert torch.equal(y2, Ys[i]) and int(idx2.item()) == idx_stream[i]
        # streamed (scan kernel) and batched (GEMM + re-rank) matchers pick the same rows; outputs equal up to the kernel
        # choice of a 1-window batch (fp32 summation order)
        # (two rows whose fp32 distances are EQUAL are a tie whatever float64 says: the two matchers sum a row's terms in different orders
        # and may then name different rows - the planted copies differ by 1e-3 of a noise vector and collapse to one distance on the
        # bf16-rounded bank now and then; such a pair must be a float64 near-tie, and its window is left out of the pose comparison)
        ib_np = ib.cpu().numpy().astype(np.int64)
        diff = np.nonzero(idx_stream != ib_np)[0]
        if len(diff):
            qs = model.encode(src[diff], mean, std)[2].reshape(len(diff), -1).double()
            rows_s, rows_b = bank.cnt_nm[idx_stream[diff]].double(), bank.cnt_nm[ib_np[diff]].double()
            if bf16:
                cen = bank.cnt_nm.double().mean(0)
                rnd = lambda t: torch.from_numpy(_bf16_round((t - cen).float().cpu().numpy())).to(t.device).double()
                rows_s, rows_b, qs = rnd(rows_s), rnd(rows_b), qs - cen
            ds, db = (qs - rows_s).norm(dim=1), (qs - rows_b).norm(dim=1)
            assert len(diff) <= 3 and bool(((ds - db).abs() <= 2e-6 * ds).all()), (diff, ds, db)
        keep = np.setdiff1d(np.arange(W), diff)
        assert float((Ys[keep] - Yb[keep]).abs().max()) < 2e-5
        # indices against the float64 search on the HIP features, outputs against the oracle on a sample
        _, _, nm_s = model.encode(src, mean, std)
        q = nm_s.cpu().numpy().reshape(W, -1)
        bank_np = bank_nm.cpu().numpy()
        if bf16:
            c = bank_np.astype(np.float64).mean(0).astype(np.float32)
            qs_np, searched = q - c, _bf16_round(bank_np - c)
        else:
            qs_np, searched = q, bank_np
        ridx, _ = O.match_bruteforce(qs_np, searched)
        mm = np.nonzero(idx_stream != ridx)[0]
        if len(mm):                                       # as above: only fp32-level ties of the planted copies on the rounded bank
            d_hip = np.linalg.norm(qs_np[mm].astype(np.float64) - searched[idx_stream[mm]].astype(np.float64), axis=1)
            d_ref = np.linalg.norm(qs_np[mm].astype(np.float64) - searched[ridx[mm]].astype(np.float64), axis=1)
            assert bf16 and len(mm) <= 3 and bool((np.abs(d_hip - d_ref) <= 2e-6 * d_ref).all()), (mm, d_hip, d_ref)
        assert len(set(idx_stream.tolist())) == W and (bf16 or np.array_equal(idx_stream, planted))
        assert len(diff) == 0 or bf16                     # fp32 rows: the copies stay 1e-3 apart, no ties
        sel = np.arange(0, W, 19)
        with torch.no_grad():
            eo, _ = O.encode(ost, src.cpu()[sel])
            Yo = O.to_mot(ost, O.decoder(ost, eo, bank_nm.view(NB, 90, 256).cpu()[torch.from_numpy(ridx[sel])])).numpy()
        assert np.abs(Ys.cpu().numpy()[sel] - Yo).max() < TOL
    # a captured step survives other work on the context: a bigger batch replaces the workspaces (generation bump) and
    # another bank becomes current in between; the next step re-captures / re-activates and stays correct
    small = ContextBank(model, bank_nm[:64].contiguous(), bank_nm[:64].view(64, 90, 256).contiguous())
    model.characterize_pair(src[:40], src[40:90], mean, std)
    small.query(bank_nm[:4].contiguous())
    y, idx = sc.step(src[3])
    assert int(idx.item()) == idx_stream[3] and torch.equal(y, Ys[3])
