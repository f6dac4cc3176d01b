"""Row N3: decoded windows -> root integration / blending / foot-lock IK -> BVH.

CPU: the oracle against what the REFERENCE'S OWN FRAME LOOP produced - tests/golden/postprocess.npz holds the results of executing
test_fullframework.py:289-641 and :643-694 from the reference's source text (tests/golden/make_golden.py: run_postprocess), for the
"Ours" stream (blend + foot-lock IK) and the context-matching "cm_" stream (neither); bvh_writer.npz is the reference writer's text.  GPU: the HIP kernels against the oracle and the fixtures.
Tolerances: heads are float32 (1e-6 absolute on unit quaternions, compared up to sign-free equality because the kernel
takes the same branch as NumPy on these inputs); the float64 clip state must agree to 1e-9."""
import os

import numpy as np
import pytest
import torch

from mocha_sigasia2023_amd import synthetic
from mocha_sigasia2023_amd.skeleton import LAYOUTS
from oracle import postprocess_oracle as P

GOLD = os.path.join(os.path.dirname(__file__), "golden")
PARENTS = np.concatenate([[-1], np.asarray(LAYOUTS["mocha"]["parents"]) + 1])


def _inputs():
    z = np.load(os.path.join(GOLD, "postprocess.npz"))
    N = int(z["seed"][1])
    Y, rvel, rang, hipvel, contact = synthetic.postprocess_inputs(int(z["seed"][0]), N)
    src_speed = np.linalg.norm(hipvel, axis=-1).mean(-1).astype(np.float32)
    return z, Y, rvel, rang, src_speed, contact


def test_oracle_matches_reference_modules():
    z, Y, rvel, rang, src_speed, contact = _inputs()
    heads, speed = P.pose_heads(Y)
    assert np.array_equal(heads[..., 3:7], z["heads_rot"])
    assert np.array_equal(speed, z["speed"])
    pos, rot, ik = P.run_clip(heads, speed, rvel, rang, src_speed, contact, PARENTS)
    assert np.abs(pos - z["pos"]).max() < 1e-12
    assert np.abs(rot - z["rot"]).max() < 1e-12
    assert np.abs(ik - z["ik_rot"]).max() < 1e-9
    assert (np.abs(ik - rot).max(axis=(1, 2)) > 1e-6).sum() > 50          # the IK really acts
    bp, be = P.bvh_channels(pos, ik)
    assert np.abs(bp - z["bvh_pos"]).max() < 1e-12
    assert np.abs(be - z["bvh_euler"]).max() < 1e-6


def test_oracle_matches_the_reference_loops_cm_stream():
    """The "cm_" stream of the same executed reference loop (test_fullframework.py:512-527, 637-641, 689-694): the decoded poses as
    they are - no blending with the previous frame, no IK - with the root integrated from the stream's own previous root."""
    z, Y, rvel, rang, src_speed, contact = _inputs()
    Ycm = synthetic.postprocess_inputs(int(z["cm_seed"][0]), int(z["cm_seed"][1]))[0]
    heads, speed = P.pose_heads(Ycm)
    assert np.array_equal(speed, z["cm_speed"])
    pos, rot, ik = P.run_clip(heads, speed, rvel, rang, src_speed, contact, PARENTS, ik_enabled=False, blend=False)
    assert np.abs(pos - z["cm_pos"]).max() < 1e-12 and np.abs(rot - z["cm_rot"]).max() < 1e-12
    assert np.array_equal(ik, rot)
    bp, be = P.bvh_channels(pos, rot)
    assert np.abs(bp - z["cm_bvh_pos"]).max() < 1e-12 and np.abs(be - z["cm_bvh_euler"]).max() < 1e-6
    assert np.abs(z["cm_pos"][1:, 1:] - heads[1:, :, 0:3]).max() == 0           # really unblended: the joints are the decoded ones


def test_oracle_exercises_every_contact_transition():
    z, Y, rvel, rang, src_speed, contact = _inputs()
    heads, speed = P.pose_heads(Y)
    pp = P.PostProcess(PARENTS)
    locks, ratios = [], []
    for i in range(len(Y)):
        pp.step(heads[i], speed[i], rvel[i], rang[i], src_speed[i], contact[i])
        locks.append([c.lock for c in pp.contacts])
        ratios.append(speed[i] / src_speed[i])
    locks = np.asarray(locks)
    assert locks.any() and not locks.all()
    # a radius-triggered unlock: lock drops while the contact label is still set
    assert ((~locks[1:]) & locks[:-1] & (contact[1:] == 1) & (contact[:-1] == 1)).any()
    assert max(ratios) > 3.0                                                # the ratio reset branch (:494-495)


def test_bvh_writer_text_equals_reference_writer(tmp_path):
    from mocha_sigasia2023_amd.postprocess import write_bvh
    z = np.load(os.path.join(GOLD, "bvh_writer.npz"))
    order = write_bvh(str(tmp_path / "o.bvh"), ["Bone%02d" % i for i in range(24)], LAYOUTS["mocha"]["parents"], z["pos"], z["rot"])
    assert open(tmp_path / "o.bvh").read() == str(z["text"])
    assert sorted(order) == list(range(24))
    with pytest.raises(ValueError):
        write_bvh(str(tmp_path / "bad.bvh"), ["a"], LAYOUTS["mocha"]["parents"], z["pos"], z["rot"])


@pytest.fixture(scope="module")
def model():
    from mocha_sigasia2023_amd import Generator, synthetic_state_dict
    return Generator(device="cuda:0").load_state_dict(synthetic_state_dict(3, 1.0)).eval()


@pytest.mark.gpu
def test_pose_heads_parity(model):
    from mocha_sigasia2023_amd.postprocess import pose_heads
    z, Y, *_ = _inputs()
    heads, speed = pose_heads(model, torch.from_numpy(Y))
    ref_h, ref_s = P.pose_heads(Y)
    assert np.abs(heads.cpu().numpy() - ref_h).max() < 2e-6
    assert np.abs(speed.cpu().numpy() - ref_s).max() < 1e-6
    # random (non-orthonormal) 6-D inputs reach all four quaternion branches; compare up to the branch's sign
    Yr = synthetic.pose_windows(8, 40)
    h2, s2 = pose_heads(model, torch.from_numpy(Yr))
    r2, rs2 = P.pose_heads(Yr)
    a, b = h2.cpu().numpy()[..., 3:7], r2[..., 3:7]
    assert np.minimum(np.abs(a - b).max(-1), np.abs(a + b).max(-1)).max() < 2e-5
    assert np.abs(np.delete(h2.cpu().numpy(), np.s_[3:7], -1) - np.delete(r2, np.s_[3:7], -1)).max() == 0
    assert np.abs(s2.cpu().numpy() - rs2).max() < 1e-6


@pytest.mark.gpu
def test_postprocess_clip_parity(model):
    from mocha_sigasia2023_amd.postprocess import PostProcessor, pose_heads
    z, Y, rvel, rang, src_speed, contact = _inputs()
    heads, speed = P.pose_heads(Y)                                           # identical float32 heads for both sides
    out = PostProcessor(model).run(heads, speed, rvel, rang, src_speed, contact)
    for k in ("pos", "rot", "ik_rot", "bvh_pos"):
        assert np.abs(out[k].cpu().numpy() - z[k]).max() < 1e-9, k
    assert np.abs(out["bvh_euler"].cpu().numpy() - z["bvh_euler"]).max() < 1e-6
    # the "cm_" stream (no blend, no IK) against the same executed reference loop
    Ycm = synthetic.postprocess_inputs(int(z["cm_seed"][0]), int(z["cm_seed"][1]))[0]
    hc, sc = P.pose_heads(Ycm)
    cm = PostProcessor(model, ik_enabled=False, blend=False).run(hc, sc, rvel, rang, src_speed, contact)
    assert np.abs(cm["pos"].cpu().numpy() - z["cm_pos"]).max() < 1e-9 and np.abs(cm["rot"].cpu().numpy() - z["cm_rot"]).max() < 1e-9
    assert torch.equal(cm["ik_rot"], cm["rot"])
    assert np.abs(cm["bvh_pos"].cpu().numpy() - z["cm_bvh_pos"]).max() < 1e-9
    assert np.abs(cm["bvh_euler"].cpu().numpy() - z["cm_bvh_euler"]).max() < 1e-6
    # device heads feeding the device loop, end to end
    dh, ds = pose_heads(model, torch.from_numpy(Y))
    out2 = PostProcessor(model).run(dh, ds, rvel, rang, src_speed, contact, bvh=False)
    assert np.abs(out2["ik_rot"].cpu().numpy() - z["ik_rot"]).max() < 1e-4
    assert "bvh_pos" not in out2


@pytest.mark.gpu
def test_postprocess_many_clips_and_options(model):
    from mocha_sigasia2023_amd.postprocess import PostProcessor
    clips = []
    for seed in (11, 12, 13):
        Y, rvel, rang, hipvel, contact = synthetic.postprocess_inputs(seed, 70)
        heads, speed = P.pose_heads(Y)
        clips.append((heads, speed, rvel, rang, np.linalg.norm(hipvel, axis=-1).mean(-1).astype(np.float32), contact))
    batch = [np.stack([c[k] for c in clips]) for k in range(6)]
    out = PostProcessor(model).run(*batch)
    for ci, c in enumerate(clips):
        pos, rot, ik = P.run_clip(*c, PARENTS)
        assert np.abs(out["pos"][ci].cpu().numpy() - pos).max() < 1e-9
        assert np.abs(out["ik_rot"][ci].cpu().numpy() - ik).max() < 1e-9
    # IK off leaves the rotations untouched; other constants are honoured
    off = PostProcessor(model, ik_enabled=False).run(*clips[0])
    assert torch.equal(off["rot"], off["ik_rot"])
    alt = PostProcessor(model, contact_bones=[24], ik_unlock_radius=0.5, dt=1.0 / 30.0).run(*clips[0][:5], clips[0][5][:, 1:])
    pos, rot, ik = P.run_clip(*clips[0][:5], clips[0][5][:, 1:], PARENTS, contact_bones=(24,), dt=1.0 / 30.0,
                              ik=dict(P.IK, unlock_radius=0.5))
    assert np.abs(alt["ik_rot"].cpu().numpy() - ik).max() < 1e-9
    with pytest.raises(RuntimeError):
        PostProcessor(model, contact_bones=[1]).run(*clips[0][:5], clips[0][5][:, 1:])     # no 4-ancestor chain
    with pytest.raises(ValueError):
        PostProcessor(model).run(clips[0][0], clips[0][1][:-1], *clips[0][2:])


@pytest.mark.gpu
def test_retarget_clip_end_to_end(model, tmp_path):
    """characterize -> heads -> clip loop -> BVH text, against the same chain evaluated by the oracles."""
    from mocha_sigasia2023_amd import ContextBank, retarget_clip, write_bvh
    from mocha_sigasia2023_amd.postprocess import pose_heads
    N = 24
    src = torch.from_numpy(synthetic.pose_windows(31, N)).cuda()
    cha = torch.from_numpy(synthetic.pose_windows(32, 40)).cuda()
    m_, s_ = synthetic.cnt_norm(7)
    mean, std = torch.from_numpy(m_).cuda(), torch.from_numpy(s_).cuda()
    enc, cnt, nm = model.encode(cha, mean, std)
    bank = ContextBank(model, nm, enc)
    _, rvel, rang, hipvel, contact = synthetic.postprocess_inputs(5, N)
    src_speed = np.linalg.norm(hipvel, axis=-1).mean(-1).astype(np.float32)
    out = retarget_clip(bank, src, mean, std, rvel, rang, src_speed, contact)
    Y = bank.characterize(src, mean, std).cpu().numpy()
    heads, speed = P.pose_heads(Y)
    pos, rot, ik = P.run_clip(heads, speed, rvel, rang, src_speed, contact, PARENTS)
    assert np.isfinite(out["bvh_euler"].cpu().numpy()).all()
    assert np.abs(out["pos"].cpu().numpy() - pos).max() < 1e-4
    assert np.abs(out["rot"].cpu().numpy() - rot).max() < 1e-4
    names = ["B%d" % i for i in range(24)]
    write_bvh(str(tmp_path / "ours.bvh"), names, LAYOUTS["mocha"]["parents"], out["bvh_pos"], out["bvh_euler"])
    text = open(tmp_path / "ours.bvh").read().splitlines()
    assert text[0] == "HIERARCHY" and "Frames: %d" % N in text and len(text[-1].split()) == 3 + 3 * 24


@pytest.mark.gpu
def test_retarget_clip_ours_end_to_end(model):
    """CVAE branch for a clip: OursSession frame loop -> heads -> clip loop, against the same chain on the oracles."""
    from mocha_sigasia2023_amd import CVAE, OursSession, retarget_clip_ours
    from mocha_sigasia2023_amd import weights as W
    from oracle import cvae_oracle as CO
    from oracle import mocha_oracle as O
    gsd = W.synthetic_state_dict(3, 1.0)
    csd = W.synthetic_cvae_state_dict(99, 1.0)
    net = CVAE(output_seq=90, latent_dim=256, depth=2, nheads=4, feedforward_dim=512, device="cuda:0").load_state_dict(csd).eval()
    rng = np.random.Generator(np.random.PCG64(9))
    stats = [(0.1 * rng.standard_normal((90, 256))).astype(np.float32), rng.uniform(0.5, 1.5, (90, 256)).astype(np.float32),
             (0.1 * rng.standard_normal((90, 256))).astype(np.float32), rng.uniform(0.5, 1.5, (90, 256)).astype(np.float32)]
    N = 6
    X = torch.from_numpy(synthetic.pose_windows(61, N + 1)).cuda()
    enc, cnt = model.encode(X)
    _, rvel, rang, hipvel, contact = synthetic.postprocess_inputs(5, N)
    src_speed = np.linalg.norm(hipvel, axis=-1).mean(-1).astype(np.float32)
    eps = torch.from_numpy(synthetic.token_features(78, 1)[0, :N].copy()).cuda()
    sess = OursSession(model, net, *stats).reset(enc[0])
    out = retarget_clip_ours(sess, enc[1:], cnt[1:], rvel, rang, src_speed, contact, eps=eps)
    # oracle chain
    tg, tc = O.to_torch_state(gsd), O.to_torch_state(csd)
    sm, ss, cm, cs = (torch.from_numpy(a) for a in stats)
    prev = enc[:1].cpu()
    Ys = []
    with torch.no_grad():
        for i in range(1, N + 1):
            cond = torch.cat([(cnt[i:i + 1].cpu() - sm) / ss, (prev - cm) / cs], dim=1)
            vae, _, _ = CO.sample(tc, cond, eps[i - 1:i].cpu())
            prev = vae * cs + cm
            Ys.append(O.to_mot(tg, O.decoder(tg, enc[i:i + 1].cpu(), prev)))
    heads, speed = P.pose_heads(torch.cat(Ys).numpy())
    pos, rot, ik = P.run_clip(heads, speed, rvel, rang, src_speed, contact, PARENTS)
    assert np.abs(out["pos"].cpu().numpy() - pos).max() < 1e-3
    assert np.abs(out["rot"].cpu().numpy() - rot).max() < 1e-3
    assert np.isfinite(out["bvh_euler"].cpu().numpy()).all()
