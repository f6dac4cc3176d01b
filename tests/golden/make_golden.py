#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by running THE REFERENCE ITSELF.

Runs only in the build container, where the reference checkout is mounted read-only at
/root/reference.  It imports the reference's ``Generator`` (model.py:15), loads the
deterministic synthetic weights of ``mocha_sigasia2023_amd.weights`` into it with
``load_state_dict(strict=True)`` (which also pins the key schema and, through the buffer
comparison below, the regenerated graph constants), runs the demo's call sequence
(test_fullframework.py:190-193,301-302) and ``Generator.forward`` (model.py:82-106) on
seeded synthetic windows, and stores inputs + outputs as ``.npz``.  The matching fixture
runs scikit-learn's ``BallTree`` exactly as the demo does (test_fullframework.py:294-296).

No reference source or bytecode is written anywhere; the fixtures are data only.

    python tests/golden/make_golden.py            # regenerates every fixture
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("MOCHA_REFERENCE", "/root/reference")
sys.dont_write_bytecode = True
sys.path.insert(0, REPO)

from mocha_sigasia2023_amd import synthetic, weights  # noqa: E402
from mocha_sigasia2023_amd.skeleton import skeleton_constants  # noqa: E402


def build_reference_generator(layout):
    """Instantiate the reference Generator; for 'mixamo' swap the two hard-coded 'mocha'
    pool modules after construction."""
    cwd = os.getcwd()
    os.chdir(REF)  # model.py:7-8 appends './net', './motion' relative to cwd
    try:
        for p in (REF, os.path.join(REF, "net"), os.path.join(REF, "etc")):
            if p not in sys.path:
                sys.path.append(p)
        from utils import get_config          # etc/utils.py:23-25
        from model import Generator           # model.py:15
        from graph import PoolJointToBodypart, UnpoolBodypartToJoint
        from transformer import mean_variance_norm
        cfg = get_config(os.path.join(REF, "configs/config.yaml"))["model"]
        if layout != "mocha":
            cfg["graph"]["joint"]["layout"] = layout
            cfg["graph"]["bodypart"]["layout"] = layout
        G = Generator(cfg)
        if layout != "mocha":
            G.mot_embedding[3] = PoolJointToBodypart(layout)
            G.to_mot[3] = UnpoolBodypartToJoint(layout)
    finally:
        os.chdir(cwd)
    return G.eval(), mean_variance_norm


def run_variant(name, layout, seed, gain, B):
    G, mvn = build_reference_generator(layout)
    ref_buffers = {k: v.numpy().copy() for k, v in G.state_dict().items()
                   if k.endswith("A_j") or k.endswith("A_b") or k in ("mot_embedding.3.weight", "to_mot.3.weight")}
    sd = weights.synthetic_state_dict(seed=seed, gain=gain, layout=layout)
    # a14: our regenerated graph constants must equal the reference's own buffers bit for bit
    for k, v in ref_buffers.items():
        assert v.shape == sd[k].shape and np.array_equal(v, sd[k]), f"graph constant mismatch: {k}"
    G.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    V = skeleton_constants(layout).V
    src = torch.from_numpy(synthetic.pose_windows(seed + 1, B, V))
    cha = torch.from_numpy(synthetic.pose_windows(seed + 2, B, V))
    out = {"src_X": src.numpy(), "cha_X": cha.numpy()}
    with torch.no_grad():
        # the demo's sequence, test_fullframework.py:190-193
        for tag, X in (("src", src), ("cha", cha)):
            tokens = G.mot_embedding(X)
            out[f"{tag}_tokens"] = tokens.numpy()
            tokens = tokens + G.pos_emb[:, :tokens.shape[1]]
            encoded = G.encoder(tokens)
            out[f"{tag}_encoded"] = encoded.numpy()
            cnt = mvn(encoded.permute(0, 2, 1)).permute(0, 2, 1)
            out[f"{tag}_cnt"] = cnt.numpy()
        # sub-stage taps (module boundaries of model.py:42-50 and :71-80)
        x = G.mot_embedding[:5](src)                       # after AvgPool2d: (B, 256, 15, 6)
        out["src_emb_pooled"] = x.numpy()
        enc_s = torch.from_numpy(out["src_encoded"])
        enc_c = torch.from_numpy(out["cha_encoded"])
        dec = G.decoder(enc_s, enc_c)                      # test_fullframework.py:301
        out["decoded"] = dec.numpy()
        out["mot_body"] = G.to_mot[:2](dec).numpy()        # after the body block: (B, 256, 15, 6)
        Y = G.to_mot(dec)                                  # :302
        out["Y"] = Y.numpy()
        Yf = G(src, cha)                                   # model.py:82-106
        assert torch.equal(Yf, Y) or float((Yf - Y).abs().max()) < 1e-6
        out["Y_forward"] = Yf.numpy()
    meta = dict(layout=layout, seed=seed, gain=gain, B=B, V=V,
                torch=torch.__version__, numpy=np.__version__)
    path = os.path.join(HERE, f"generator_{name}.npz")
    np.savez(path, **out, meta=np.array(repr(meta)))
    print(name, {k: (v.shape, float(np.abs(v).max())) for k, v in out.items()})
    return path


def run_graph_constants():
    """Reference buffers for both layouts (row a14), stored so the CPU test needs no reference."""
    out = {}
    for layout in ("mocha", "mixamo"):
        G, _ = build_reference_generator(layout)
        sd = G.state_dict()
        out[f"{layout}_A_j"] = sd["mot_embedding.2.A_j"].numpy()
        out[f"{layout}_A_b"] = sd["mot_embedding.5.A_b"].numpy()
        out[f"{layout}_pool"] = sd["mot_embedding.3.weight"].numpy()
        out[f"{layout}_unpool"] = sd["to_mot.3.weight"].numpy()
    np.savez(os.path.join(HERE, "graph_constants.npz"), **out)


def run_match():
    """BallTree k=1 exactly as test_fullframework.py:293-296,440-443, on seeded synthetic cnt."""
    import sklearn
    from sklearn.neighbors import BallTree
    out = {}
    mean, std = synthetic.cnt_norm(90)
    for nb, q, seed in ((64, 16, 100), (585, 16, 200)):
        cha_cnt = synthetic.token_features(seed, nb)
        src_cnt = synthetic.token_features(seed + 1, q)
        # make a few queries near-duplicates of bank rows so the answer is not arbitrary
        src_cnt[:4] = cha_cnt[[3, nb - 1, nb // 2, 7]] + 0.05 * src_cnt[:4]
        cha_nm = (cha_cnt - mean[np.newaxis]) / std[np.newaxis]
        tree = BallTree(cha_nm.reshape(cha_nm.shape[0], -1))
        src_nm = (src_cnt - mean[np.newaxis]) / std[np.newaxis]
        dist, idx = tree.query(src_nm.reshape(src_nm.shape[0], -1), k=1, return_distance=True)
        out[f"n{nb}_idx"] = idx[:, 0].astype(np.int64)
        out[f"n{nb}_dist"] = dist[:, 0].astype(np.float64)
        out[f"n{nb}_seed"] = np.array([seed, nb, q])
    out["meta"] = np.array(repr(dict(sklearn=sklearn.__version__, cnt_norm_seed=90)))
    np.savez(os.path.join(HERE, "match_balltree.npz"), **out)
    print("match", {k: v for k, v in out.items() if k.endswith("idx")})


def run_cvae():
    """CVAE.sample(c, deterministic=True) of the reference (model_CVAE.py:44-46) with synthetic weights,
    constructed as the demo does (test_fullframework.py:52-58)."""
    cwd = os.getcwd(); os.chdir(REF)
    try:
        if REF not in sys.path:
            sys.path.append(REF)
        import torch.nn.functional as F
        from model_CVAE import CVAE
        net = CVAE(output_seq=90, latent_dim=256, depth=2, nheads=4, feedforward_dim=512, dropout=0.1, activation=F.relu).eval()
    finally:
        os.chdir(cwd)
    sd = weights.synthetic_cvae_state_dict(seed=99, gain=1.0)
    ref_sd = net.state_dict()
    for k, v in sd.items():
        assert tuple(ref_sd[k].shape) == v.shape, k
    missing, unexpected = net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    assert not unexpected and all(k.startswith("encoder.") or k.endswith("pos_encoder.pe") for k in missing), missing
    assert np.array_equal(ref_sd["prior_net.pos_encoder.pe"][0, :182].numpy(), weights.sincos_pe(182))
    c = torch.from_numpy(synthetic.token_features(500, 2 * 2).reshape(2, 180, 256))
    with torch.no_grad():
        mu, logvar = net.prior(c)
        out = net.sample(c, deterministic=True)
    np.savez(os.path.join(HERE, "cvae_sample.npz"), c=c.numpy(), mu=mu.numpy(), logvar=logvar.numpy(), out=out.numpy(),
             meta=np.array(repr(dict(seed=99, gain=1.0, torch=torch.__version__))))
    print("cvae", {k: float(np.abs(v).max()) for k, v in dict(mu=mu.numpy(), logvar=logvar.numpy(), out=out.numpy()).items()})


def run_featurize():
    """test_fullframework.py:141-185 executed with the reference's own quaternion library (motion/quat.py)
    on synthetic local bone features; only the X features are stored."""
    sys.path.insert(0, os.path.join(REF, "motion"))
    import quat                                           # the reference's motion/quat.py
    from mocha_sigasia2023_amd.skeleton import LAYOUTS
    parents = np.concatenate([[-1], np.asarray(LAYOUTS["mocha"]["parents"]) + 1])   # :101-102
    window = 60
    Yrot, Ypos, Yvel, Yang = synthetic.bone_windows(321, 3)
    Grot, Gpos, Gvel, Gang = quat.fk_vel(Yrot, Ypos, Yvel, Yang, parents)
    Gpos[:, :, 0:1] = np.repeat(Gpos[:, -1:, 0:1], window, axis=1)
    Grot[:, :, 0:1] = np.repeat(Grot[:, -1:, 0:1], window, axis=1)
    Gvel[:, :, 0:1] = np.repeat(Gvel[:, -1:, 0:1], window, axis=1)
    Gang[:, :, 0:1] = np.repeat(Gang[:, -1:, 0:1], window, axis=1)
    Xpos = quat.inv_mul_vec(Grot[:, :, 0:1], Gpos - Gpos[:, :, 0:1])
    Xrot = quat.inv_mul(Grot[:, :, 0:1], Grot)
    Xtxy = quat.to_xform_xy(Xrot).astype(np.float32)
    Xvel = quat.inv_mul_vec(Grot[:, :, 0:1], Gvel)
    Xang = quat.inv_mul_vec(Grot[:, :, 0:1], Gang)
    b, ns, nj, _, _ = Xtxy.shape
    X = np.concatenate([Xpos, Xtxy.reshape(b, ns, nj, -1), Xvel, Xang], axis=-1)
    np.savez(os.path.join(HERE, "featurize.npz"), X=X.astype(np.float32), seed=np.array([321, 3]))
    print("featurize", X.shape, X.dtype, float(np.abs(X).max()))


def run_database():
    """Row N4's tail: a small synthetic ``database.bin`` in the byte layout preprocess/generate_database_bin.py:228-246 writes,
    read back with THE REFERENCE's ``load_database`` (etc/utils.py:144-190); the step-1 windowing, FK / re-rooting and z-score
    by EXECUTING lines 93-165 of the reference's collect_CVAE_feature_action.py (its ``main`` is not importable piecewise) in a
    namespace prepared with that file's own variable names.  Stored: the file bytes, what the reference reader returns, the
    reference's window ranges / labels / first frames, and its normalised features X for a few windows."""
    import io
    import struct
    import textwrap
    sys.path.insert(0, os.path.join(REF, "motion")); sys.path.insert(0, os.path.join(REF, "etc"))
    import quat
    from utils import load_database as ref_load_database
    from mocha_sigasia2023_amd.skeleton import LAYOUTS
    r = np.random.Generator(np.random.PCG64(2024))
    lengths, styles, actions = [75, 64, 90, 61, 70], [2, 2, 5, 2, 2], [1, 3, 1, 4, 2]
    nframes, nbones = sum(lengths), 25
    rot = r.standard_normal((nframes, nbones, 4)).astype(np.float32)
    rot /= np.sqrt((rot * rot).sum(-1, keepdims=True)); rot = np.where(rot[..., :1] > 0, rot, -rot).astype(np.float32)
    db = {"bone_positions": (0.3 * r.standard_normal((nframes, nbones, 3))).astype(np.float32),
          "bone_velocities": r.standard_normal((nframes, nbones, 3)).astype(np.float32), "bone_rotations": rot,
          "bone_angular_velocities": r.standard_normal((nframes, nbones, 3)).astype(np.float32),
          "bone_parents": np.concatenate([[-1], np.asarray(LAYOUTS["mocha"]["parents"]) + 1]).astype(np.int32),
          "range_starts": np.cumsum([0] + lengths[:-1]).astype(np.int32), "range_stops": np.cumsum(lengths).astype(np.int32),
          "style_labels": np.asarray(styles, np.int32), "action_labels": np.asarray(actions, np.int32),
          "contact_states": (r.uniform(size=(nframes, 2)) > 0.5).astype(np.uint8)}
    buf = io.BytesIO()                                    # preprocess/generate_database_bin.py:235-246, same order and headers
    for k in ("bone_positions", "bone_velocities", "bone_rotations", "bone_angular_velocities"):
        buf.write(struct.pack("II", nframes, nbones) + db[k].ravel().tobytes())
    buf.write(struct.pack("I", nbones) + db["bone_parents"].ravel().tobytes())
    for k in ("range_starts", "range_stops", "style_labels", "action_labels"):
        buf.write(struct.pack("I", len(lengths)) + db[k].ravel().tobytes())
    buf.write(struct.pack("II", nframes, 2) + db["contact_states"].ravel().tobytes())
    raw = buf.getvalue()
    tmp = os.path.join("/tmp", "mocha_golden_database.bin")
    open(tmp, "wb").write(raw)
    database = ref_load_database(tmp)                     # the reference reader
    out = {"bin": np.frombuffer(raw, dtype=np.uint8)}
    import hashlib
    for k, v in database.items():                         # what the reference reader returned: small arrays verbatim, the frame arrays
        v = np.asarray(v)                                 # (already in `bin`) as shape + dtype + SHA-256 of their bytes
        if v.size <= 64:
            out["db_" + k] = v
        else:
            out["dbsha_" + k] = np.array([str(v.shape), str(v.dtype), hashlib.sha256(np.ascontiguousarray(v).tobytes()).hexdigest()])
    # ---- the reference's windowing + featurisation, executed from its own source text
    src = open(os.path.join(REF, "collect_CVAE_feature_action.py")).read().splitlines()
    code = textwrap.dedent("\n".join(src[92:165]))        # lines 93-165
    Xm = r.standard_normal((1, 1, 25, 15)).astype(np.float32); Xs = r.uniform(0.5, 2.0, (1, 1, 25, 15)).astype(np.float32)
    ns = {"np": np, "quat": quat, "parents": database["bone_parents"], "contacts": database["contact_states"],
          "range_starts": database["range_starts"], "range_stops": database["range_stops"], "style_labels": database["style_labels"],
          "action_labels": database["content_labels"],    # the reader's name for them (etc/utils.py:172-173)
          "Ypos": database["bone_positions"].astype(np.float32), "Yrot": database["bone_rotations"].astype(np.float32),
          "Yvel": database["bone_velocities"].astype(np.float32), "Yang": database["bone_angular_velocities"].astype(np.float32),
          "target_style_label": [2], "target_action_label": [1, 3, 4], "style_names": ["s%d" % i for i in range(8)],
          "action_names": ["a%d" % i for i in range(8)], "X_mean": Xm, "X_std": Xs, "print": lambda *a, **k: None}
    exec(compile(code, "collect_CVAE_feature_action.py[93:165]", "exec"), ns)
    N = len(ns["action_label"])
    # first frame of every window: located by matching the reference's slices in the frame arrays
    pos = database["bone_positions"]
    starts = np.array([next(g for g in range(nframes - 59) if np.array_equal(pos[g:g + 60], ns["Ypos"][k])) for k in range(N)], np.int64)
    sel = np.array([0, 7, N // 2, N - 1])
    out.update(range_starts=ns["cha_range_starts"], range_stops=ns["cha_range_stops"], action_label=ns["action_label"], starts=starts,
               X_sel=ns["X"][sel].astype(np.float32), sel=sel, X_mean=Xm, X_std=Xs, style_keep=np.array([2]), action_keep=np.array([1, 3, 4]))
    np.savez_compressed(os.path.join(HERE, "database_bank.npz"), **out)
    os.remove(tmp)
    print("database", {k: getattr(v, "shape", v) for k, v in out.items() if not k.startswith("db_")})


def run_postprocess():
    """The demo's per-frame post-processing (test_fullframework.py:303-437 for the first frame, :457-632 after it)
    driven through the reference's own motion/quat.py and motion/Inertialization.py on synthetic decoded windows."""
    cwd = os.getcwd(); os.chdir(REF)
    try:
        sys.path.insert(0, os.path.join(REF, "motion"))
        import quat
        import Inertialization as inert
    finally:
        os.chdir(cwd)
    from mocha_sigasia2023_amd.skeleton import LAYOUTS
    par = np.concatenate([[-1], np.asarray(LAYOUTS["mocha"]["parents"]) + 1])
    feet = np.array([5, 24]); dt = 1.0 / 60.0
    buf, foot_h, radius, half = 0.015, 0.02, 0.2, 0.1
    N = 120
    Y, rvel, rang, hipvel, contact = synthetic.postprocess_inputs(77, N)
    ident = np.array([1, 0, 0, 0])
    P, R, RIK = [], [], []
    heads_rot, speeds = [], []
    for i in range(N):
        W = Y[i]
        jp, jv, ja = W[-1, :, :3], W[..., 9:12], W[-1, :, 12:15]
        jr = quat.from_xform_xy(W[-1, :, 3:9].reshape(jp.shape[0], 3, 2))
        heads_rot.append(jr)
        ratio = np.linalg.norm(jv[:, 0], axis=1).mean() / np.linalg.norm(hipvel[i], axis=1).mean()
        speeds.append(np.linalg.norm(jv[:, 0], axis=1).mean())
        if ratio > 3.0 or ratio < 0.33:
            ratio = 1.0
        q0, p0 = (ident, np.array([0, 0, 0])) if i == 0 else (R[-1][0], P[-1][0])
        wv = quat.mul_vec(q0, rvel[i] * ratio); wa = quat.mul_vec(q0, rang[i])
        pos = np.concatenate([(p0 + wv * dt)[None], jp]); vel = np.concatenate([wv[None], jv[-1]])
        rot = np.concatenate([quat.mul(q0, quat.from_scaled_angle_axis(wa * dt))[None], jr]); ang = np.concatenate([wa[None], ja])
        if i == 0:
            nc = feet.size
            st = dict(state=np.zeros(nc, bool), lock=np.zeros(nc, bool), pos=np.zeros((nc, 3)), vel=np.zeros((nc, 3)),
                      point=np.zeros((nc, 3)), target=np.zeros((nc, 3)), ox=np.zeros((nc, 3)), ov=np.zeros((nc, 3)))
            for b in range(nc):
                bp, bv, _, _ = quat.fk_vel_bone(pos, vel, rot, ang, par, feet[b])
                st["pos"][b] = bp; st["vel"][b] = bv; st["point"][b] = bp; st["target"][b] = bp
            gpos, grot, done = np.zeros((len(par), 3)), np.zeros((len(par), 4)), np.zeros(len(par), bool)
            P.append(pos); R.append(rot); RIK.append(rot)
            continue
        lpos = ((P[-1] + vel * dt) * 0.5 + pos * 0.5).copy()
        arot = rot.copy()
        for b in range(feet.size):
            toe = feet[b]; heel = par[toe]; knee = par[heel]; hip = par[knee]; up = par[hip]
            done[:] = False
            quat.fk_partial(gpos, grot, done, lpos, rot, par, toe)
            (st["state"][b], st["lock"][b], st["pos"][b], st["vel"][b], st["point"][b], st["target"][b], st["ox"][b], st["ov"][b]) = \
                inert.contact_update(st["state"][b], st["lock"][b], st["pos"][b], st["vel"][b], st["point"][b], st["target"][b],
                                     st["ox"][b], st["ov"][b], gpos[toe], bool(contact[i, b]), radius, foot_h, half, dt)
            clamp = st["pos"][b]
            clamp[1] = np.max([clamp[1], foot_h])
            arot[hip], arot[knee] = quat.ik_two_bone(arot[hip], arot[knee], gpos[hip], gpos[knee], gpos[heel],
                                                     clamp + (gpos[heel] - gpos[toe]),
                                                     quat.mul_vec(grot[knee], np.array([0.0, 1.0, 0.0], dtype=np.float32)),
                                                     grot[hip], grot[knee], grot[up], buf)
        P.append(lpos); R.append(rot); RIK.append(arot)
    P, R, RIK = np.stack(P), np.stack(R), np.stack(RIK)
    grot_all, gpos_all = quat.fk(RIK, P, par)                           # root merge of :677-681 and Euler channels of :697
    bp = P[:, 1:].copy(); bp[:, 0] = gpos_all[:, 1]
    br = RIK[:, 1:].copy(); br[:, 0] = grot_all[:, 1]
    np.savez(os.path.join(HERE, "postprocess.npz"), seed=np.array([77, N]), pos=P, rot=R, ik_rot=RIK,
             heads_rot=np.stack(heads_rot).astype(np.float32), speed=np.asarray(speeds, np.float32),
             bvh_pos=bp, bvh_euler=np.degrees(quat.to_euler(br)))
    print("postprocess", P.shape, float(np.abs(P).max()), "ik changed rotations on",
          int((np.abs(RIK - R).max(axis=(1, 2)) > 1e-9).sum()), "frames")


def run_bvh():
    """The reference writer (motion/bvh.py:179-224) on a small seeded animation; the produced text is the expected output."""
    import tempfile
    sys.path.insert(0, os.path.join(REF, "motion"))
    import bvh
    from mocha_sigasia2023_amd.skeleton import LAYOUTS
    parents = np.asarray(LAYOUTS["mocha"]["parents"])
    r = np.random.Generator(np.random.PCG64(5))
    V, N = len(parents), 4
    pos = r.standard_normal((N, V, 3)); rot = 90.0 * r.standard_normal((N, V, 3))
    names = ["Bone%02d" % i for i in range(V)]
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "o.bvh")
        bvh.save(path, {"rotations": rot, "positions": pos, "offsets": pos[0], "parents": parents, "names": names, "order": "zyx"})
        text = open(path).read()
    np.savez(os.path.join(HERE, "bvh_writer.npz"), pos=pos, rot=rot, text=np.array(text))
    print("bvh", len(text), "chars")


if __name__ == "__main__":
    torch.manual_seed(0)
    run_graph_constants()
    run_variant("mocha24_g1", "mocha", seed=1777, gain=1.0, B=2)
    run_variant("mocha24_g2", "mocha", seed=4242, gain=2.0, B=1)
    run_variant("mixamo22_g1", "mixamo", seed=2222, gain=1.0, B=1)
    run_match()
    run_cvae()
    run_featurize()
    run_database()
    run_postprocess()
    run_bvh()
