"""Every kernel class of the library as a VICTIM beside the aggressor that corrupted mocha_body_front in round 2 (run with -m gpu).

Round 2 found one kernel whose results depended on what shared its CU: a build of mocha_body_front with compiler-formed
v_pk_fma_f32 (op_sel on a high register) intermittently produced zeros in lanes 48-63 while ANOTHER STREAM ran kernels that
issue ordinary VALU instructions between bf16 MFMAs (mocha_gemm_x3's hand-interleaved plane split; profiles/r02/
f_body_front_repro.txt).  The fix removed packed fp32 from that kernel; what the other kernels do under the same aggressor was a
guess.  This test replaces the guess: the aggressor - a second context streaming large-batch encoder passes (mocha_gemm_x3 and
mocha_attention_x3 back to back) on its own stream - runs while each victim call is repeated on another stream (same and
different priority = different hardware queues), and every repetition is compared BIT FOR BIT with the victim's solo result.
Victims cover every .hip translation unit: embedding front ends, window sums, body/joint adjacency kernels, instance norm /
AdaIN, the three GEMM engines at the sizes that select them, both attention engines, the streaming matcher for 1..8 queries
(fp32 and bf16 banks; its inner loop is v_pk_fma_f32 by design), the many-query matcher with its select kernel, top-k, gathers,
the final projection, featurisation, pose heads / post-processing and the CVAE sampler.
include/mocha_hip.h cites this test for its statement about caller-side streams.
"""
import numpy as np
import pytest
import torch

from mocha_sigasia2023_amd import CVAE, ContextBank, Generator, synthetic, weights

pytestmark = pytest.mark.gpu
REPS = 1000            # repetitions per victim, split over the stream-priority settings


def _bitwise(a, b):
    if isinstance(a, (tuple, list)):
        return all(_bitwise(x, y) for x, y in zip(a, b))
    if a.dtype.is_floating_point:                   # NaNs (none are expected) must not hide a difference
        return torch.equal(a.view(torch.int32 if a.dtype == torch.float32 else torch.int64), b.view(torch.int32 if b.dtype == torch.float32 else torch.int64))
    return torch.equal(a, b)


def _clone(r):
    return tuple(x.clone() for x in r) if isinstance(r, (tuple, list)) else r.clone()


class Aggressor:
    """Keeps the chip busy with plane GEMMs and plane attention from its own context and stream."""

    def __init__(self):
        self.model = Generator(device="cuda:0").load_state_dict(weights.synthetic_state_dict(77, 1.0)).eval()
        self.tokens = torch.from_numpy(synthetic.token_features(1, 585)).cuda()
        self.model.encoder(self.tokens)
        torch.cuda.synchronize()

    def enqueue(self, stream, n):
        with torch.cuda.stream(stream):
            for _ in range(n):
                self.model.encoder(self.tokens)             # 2 x (qkv GEMM, attention, out-proj, FF1, FF2): ~2 ms per call


def victims():
    """name -> zero-argument callable returning a tensor or tuple of tensors; everything it needs is created here, once."""
    dev = "cuda:0"
    sd = weights.synthetic_state_dict(5, 1.3)
    m = Generator(device=dev).load_state_dict(sd).eval()
    mean, std = (torch.from_numpy(a).cuda() for a in synthetic.cnt_norm(3))
    out = {}
    # -- network stages at the batch sizes that select each GEMM kernel (skinny / 64x64 f32 / plane engine) and both attentions
    for B in (1, 24, 160):
        X = torch.from_numpy(synthetic.pose_windows(10 + B, B)).cuda()
        tok = torch.from_numpy(synthetic.token_features(20 + B, B)).cuda()
        cha = torch.from_numpy(synthetic.token_features(30 + B, B)).cuda()
        out[f"mot_embedding[{B}]"] = lambda X=X: m.mot_embedding(X)
        out[f"encoder[{B}]"] = lambda tok=tok: m.encoder(tok)
        out[f"decoder[{B}]"] = lambda tok=tok, cha=cha: m.decoder(tok, cha)
        out[f"to_mot[{B}]"] = lambda tok=tok: m.to_mot(tok)
        out[f"encode+mvn[{B}]"] = lambda X=X: m.encode(X, mean, std)
    # -- the opt-in two-plane fp16 GEMM engine (gemm_h2.hip: VALU between fp16 MFMAs, per-window bound vectors, atomics in the epilogue)
    mh = Generator(device=dev).load_state_dict(sd).eval()
    mh.set_option("gemm_f16x2", 1)
    Xh = torch.from_numpy(synthetic.pose_windows(170, 160)).cuda()
    tokh = torch.from_numpy(synthetic.token_features(180, 160)).cuda()
    chah = torch.from_numpy(synthetic.token_features(190, 160)).cuda()
    out["mot_embedding[160], f16x2"] = lambda: mh.mot_embedding(Xh)
    out["encoder[160], f16x2"] = lambda: mh.encoder(tokh)
    out["decoder[160], f16x2"] = lambda: mh.decoder(tokh, chah)
    out["to_mot[160], f16x2"] = lambda: mh.to_mot(tokh)
    out["forward[160], f16x2"] = lambda: mh(Xh, Xh)
    m32 = Generator(device=dev).load_state_dict(sd).eval()
    m32.set_option("gemm_bf16x3", 0).set_option("attention_bf16x3", 0)
    X32 = torch.from_numpy(synthetic.pose_windows(41, 96)).cuda()
    out["forward[96], exact-f32 engines"] = lambda: m32(X32, X32)
    # -- matchers
    r = np.random.Generator(np.random.PCG64(9))
    bank_nm = torch.from_numpy(r.standard_normal((1500, 90 * 256)).astype(np.float32)).cuda()
    bank_enc = bank_nm.view(1500, 90, 256)
    q = torch.from_numpy(r.standard_normal((40, 90 * 256)).astype(np.float32)).cuda()
    for bf16 in (False, True):
        mm = Generator(device=dev).load_state_dict(sd).eval()
        bank = ContextBank(mm, bank_nm, bank_enc, bf16=bf16)
        tag = "bf16" if bf16 else "f32"
        for Q in (1, 2, 3, 4, 8):
            out[f"match_stream[{Q}, {tag}]"] = lambda bank=bank, Q=Q: bank.query(q[:Q])
        out[f"match many[40, {tag}]"] = lambda bank=bank: bank.query(q)
        out[f"match_topk[5 x k=4, {tag}]"] = lambda bank=bank: bank.query(q[:5], k=4)
        out[f"gather+blend[{tag}]"] = lambda bank=bank: (bank.gather(torch.tensor([3, 1499, 0], dtype=torch.int32)),
                                                         bank.gather_blend(*reversed(bank.query(q[:5], k=4)), temperature=2.0))
        Xs = torch.from_numpy(synthetic.pose_windows(51, 6)).cuda()
        out[f"characterize[6, {tag}]"] = lambda bank=bank, Xs=Xs: bank.characterize(Xs, mean, std, return_index=True)
    # -- pair step (bank packing, centroid, row norms, plane-engine coarse pass, select)
    Xp, Cp = torch.from_numpy(synthetic.pose_windows(61, 90)).cuda(), torch.from_numpy(synthetic.pose_windows(62, 70)).cuda()
    out["characterize_pair[90 + 70]"] = lambda: m.characterize_pair(Xp, Cp, mean, std, return_index=True)
    # -- featurisation / pose heads / post-processing / CVAE
    J = m.V + 1
    rot = torch.from_numpy(r.standard_normal((4, 60, J, 4)).astype(np.float32)).cuda()
    rot = rot / rot.norm(dim=-1, keepdim=True)
    pos, vel, ang = (torch.from_numpy(r.standard_normal((4, 60, J, 3)).astype(np.float32)).cuda() for _ in range(3))
    out["featurize[4]"] = lambda: m.featurize(rot, pos, vel, ang)
    from mocha_sigasia2023_amd import postprocess as P
    Y = torch.from_numpy(r.standard_normal((12, 60, m.V, 15)).astype(np.float32)).cuda()
    out["pose_heads[12]"] = lambda: P.pose_heads(m, Y)
    cv = CVAE(device=dev).load_state_dict(weights.synthetic_cvae_state_dict(3)).eval()
    cond = torch.from_numpy(r.standard_normal((2, 180, 256)).astype(np.float32)).cuda()
    out["cvae.sample[2]"] = lambda: cv.sample(cond, deterministic=True)
    return out


def _run_beside(agg, fn, solo, reps, prios=(0, -1), burst=6, per_burst=20):
    """Repeat `fn` on a victim stream while the aggressor's work is in flight on another; count repetitions that differ from `solo`."""
    s_agg = torch.cuda.Stream()
    bad = {}
    for prio in prios:
        s_vic = torch.cuda.Stream(priority=prio)
        n_bad = done = 0
        while done < reps:
            agg.enqueue(s_agg, burst)                    # ~2 ms of plane GEMMs / attention per call in flight
            batch = []
            with torch.cuda.stream(s_vic):
                for _ in range(per_burst):
                    batch.append(_clone(fn()))
            torch.cuda.synchronize()
            n_bad += sum(0 if _bitwise(b, solo) else 1 for b in batch)
            done += len(batch)
        bad[prio] = (n_bad, done)
    return bad


def test_harness_detects_the_round2_canary():
    """Sensitivity of this harness: the round-2 build of mocha_body_front (tests/canary/canary.hip: LDS-fed coefficients, packed
    fp32 with op_sel on a high register) must FAIL under it, exactly as it did in the pipeline; the shipped kernel, same input and
    same harness, must not.  Without this the test below would prove nothing."""
    import ctypes as C
    import os
    import subprocess
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "canary")
    lib_path = os.path.join(here, "libmocha_canary.so")
    if not os.path.exists(lib_path) or os.path.getmtime(lib_path) < os.path.getmtime(os.path.join(here, "canary.hip")):
        subprocess.run(["make", "-C", here], check=True)
    lib = C.CDLL(lib_path)
    lib.canary_body_front_launch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    agg = Aggressor()
    frames = 300 * 15                                           # 300 windows, the size of the round-2 reproduction
    x = torch.from_numpy(synthetic.token_features(5, 300)).cuda()
    Ab = torch.from_numpy(weights.synthetic_state_dict(5, 1.0)["mot_embedding.5.A_b"]).cuda().contiguous()
    out = torch.empty((frames * 6, 512), dtype=torch.float32, device="cuda:0")

    def canary():
        rc = lib.canary_body_front_launch(x.data_ptr(), Ab.data_ptr(), out.data_ptr(), frames, torch.cuda.current_stream().cuda_stream)
        assert rc == 0
        return out
    canary(); torch.cuda.synchronize()
    solo = out.clone()
    for _ in range(50):                                         # alone on the chip it is deterministic
        canary(); torch.cuda.synchronize()
        assert _bitwise(out, solo)
    bad = _run_beside(agg, canary, solo, 1500)
    total_bad = sum(b for b, _ in bad.values())
    print(f"canary: repetitions differing from the solo result by victim-stream priority: {bad}")
    assert total_bad > 0, f"the harness no longer reproduces the round-2 failure with the canary kernel: {bad}"
    # the shipped kernel on the same input, through the library (to_mot's first stage is mocha_body_front on the tokens)
    m = Generator(device="cuda:0").load_state_dict(weights.synthetic_state_dict(5, 1.0)).eval()
    fn = lambda: m.to_mot(x)
    fn(); torch.cuda.synchronize()
    solo2 = fn().clone(); torch.cuda.synchronize()
    bad2 = _run_beside(agg, fn, solo2, 400)
    assert sum(b for b, _ in bad2.values()) == 0, bad2


def test_every_kernel_is_bit_stable_beside_the_plane_gemm_aggressor():
    agg = Aggressor()
    vic = victims()
    solo = {}
    for name, fn in vic.items():
        fn(); torch.cuda.synchronize()                   # warm-up (lazy allocations, packed images)
        solo[name] = _clone(fn())
        torch.cuda.synchronize()
        again = fn(); torch.cuda.synchronize()
        assert _bitwise(again, solo[name]), f"{name}: not even run-to-run deterministic on an idle chip"
    failures = {}
    for name, fn in vic.items():
        for prio, (bad, done) in _run_beside(agg, fn, solo[name], REPS // 2).items():
            if bad:
                failures[f"{name} (victim stream priority {prio})"] = f"{bad} of {done} repetitions differ from the solo result"
    assert not failures, failures
