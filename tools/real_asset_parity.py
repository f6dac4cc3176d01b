#!/usr/bin/env python3
"""Real-asset parity in one command - for the day `gen_125.pt`, `norm.npz` and `cnt_norm.npz` of
download.sh:3-25 exist.  Two halves, because the reference never travels to the GPU box:

  build container (imports /root/reference):
      python tools/real_asset_parity.py make --ckpt gen_125.pt --norm norm.npz [--cnt-norm cnt_norm.npz] --out real_fixture.npz
    loads the checkpoint with the reference's own reader semantics (trainer.py:224-247: torch.load(...)['gen_ema'], `module.`
    prefix accepted), runs the demo's call sequence (test_fullframework.py:186-194, 293-303) on a seeded window set and writes,
    as DATA only: the windows, every stage's output (tokens, encoded, cnt, z-scored cnt, BallTree k=1 indices, decoded, Y, the
    de-normalised Y), per decoder layer the AdaIN gains gamma of net/transformer.py:108-113 as a histogram of |1 + gamma| and its
    minimum (how ill-conditioned the instance norm behind AdaIN is with trained weights: DESIGN.md section 2), the reference's
    own batch-1-against-batch-all self-consistency, and the checkpoint file's sha256.

  GPU box (C ABI only, no reference):
      python tools/real_asset_parity.py replay --ckpt gen_125.pt --fixture real_fixture.npz [--norm norm.npz] [--cnt-norm cnt_norm.npz]
    loads the SAME file through mocha_sigasia2023_amd.Generator.load_state_dict, replays every stage from the fixture's inputs and
    prints max |hip - ref| per stage against 1e-4 (exit status 1 if the de-normalised / normalised Y exceeds it on a stage the
    reference reproduces itself to better than that).

A fixture of tests/golden/ (generator_*.npz: same key names, no matching step) replays too - that is how tests/test_real_asset_tool.py
validates the GPU half today, on a checkpoint file in the reference Trainer's schema with synthetic weights.
"""
import argparse
import hashlib
import json
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
TOL = 1e-4
GAMMA_BINS = [0.0, 1e-3, 3e-3, 1e-2, 3e-2, 0.1, 0.3, 1.0, 3.0, 1e30]      # histogram edges of |1 + gamma|


def sha256_of(path):
    h = hashlib.sha256()
    with open(path, "rb") as f:
        for blk in iter(lambda: f.read(1 << 20), b""):
            h.update(blk)
    return h.hexdigest()


def load_norms(norm_path, cnt_norm_path, V, C=15):
    """(X_mean, X_std, Y_mean, Y_std) with the root bone first, shape (V + 1, C) - test_fullframework.py:64-70 - or Nones;
    (cnt_mean, cnt_std) of shape (90, 256) - :73-75 - or (0, 1)."""
    pose = [None] * 4
    if norm_path:
        z = np.load(norm_path, allow_pickle=True)
        pose = [np.asarray(z[k], np.float32).reshape(V + 1, C) for k in ("X_mean", "X_std", "Y_mean", "Y_std")]
    if cnt_norm_path:
        z = np.load(cnt_norm_path, allow_pickle=True)
        cnt = (np.asarray(z["mean"], np.float32).reshape(90, 256), np.asarray(z["std"], np.float32).reshape(90, 256))
    else:
        cnt = (np.zeros((90, 256), np.float32), np.ones((90, 256), np.float32))
    return pose, cnt


def strip_module(sd):
    return {(k[len("module."):] if k.startswith("module.") else k): v for k, v in sd.items()}


# ------------------------------------------------------------------------------------------------ build-container half
def make(a):
    import torch
    from mocha_sigasia2023_amd import synthetic
    from mocha_sigasia2023_amd.skeleton import skeleton_constants
    sys.path.insert(0, os.path.join(REPO, "tests", "golden"))
    import make_golden as MG                                  # build_reference_generator: the reference's own Generator (model.py:15)
    from sklearn.neighbors import BallTree                    # test_fullframework.py:6

    layout = a.layout
    V = skeleton_constants(layout).V
    G, mvn = MG.build_reference_generator(layout)
    ck = torch.load(a.ckpt, map_location="cpu")
    state = strip_module(ck["gen_ema"] if isinstance(ck, dict) and "gen_ema" in ck else ck)        # trainer.py:239-240
    G.load_state_dict(state, strict=True)
    (X_mean, X_std, Y_mean, Y_std), (cnt_mean, cnt_std) = load_norms(a.norm, a.cnt_norm, V)
    B = a.windows
    src = torch.from_numpy(synthetic.pose_windows(a.seed + 1, B, V))      # z-scored poses: what X is after test_fullframework.py:186
    cha = torch.from_numpy(synthetic.pose_windows(a.seed + 2, B, V))
    out = {"src_X": src.numpy(), "cha_X": cha.numpy()}
    with torch.no_grad():
        for tag, X in (("src", src), ("cha", cha)):                       # :188-193
            tokens = G.mot_embedding(X)
            out[f"{tag}_tokens"] = tokens.numpy()
            encoded = G.encoder(tokens + G.pos_emb[:, :tokens.shape[1]])
            out[f"{tag}_encoded"] = encoded.numpy()
            out[f"{tag}_cnt"] = mvn(encoded.permute(0, 2, 1)).permute(0, 2, 1).numpy()
        nm = lambda c: (c - cnt_mean[None]) / cnt_std[None]               # :293-295
        cha_nm, src_nm = nm(out["cha_cnt"]), nm(out["src_cnt"])
        out["cha_cnt_nm"], out["src_cnt_nm"] = cha_nm, src_nm
        tree = BallTree(cha_nm.reshape(B, -1))
        dist, idx = tree.query(src_nm.reshape(B, -1), k=1, return_distance=True)
        out["frame_index"], out["frame_dist"] = idx[:, 0].astype(np.int32), dist[:, 0]
        enc_s = torch.from_numpy(out["src_encoded"])
        enc_c = torch.from_numpy(out["cha_encoded"])[torch.from_numpy(idx[:, 0])]
        dec = G.decoder(enc_s, enc_c)                                     # :301
        out["decoded"] = dec.numpy()
        Y = G.to_mot(dec)                                                 # :302
        out["Y"] = Y.numpy()
        if Y_mean is not None:
            out["Y_denorm"] = (Y.numpy() * Y_std[None, None, 1:] + Y_mean[None, None, 1:]).astype(np.float32)      # :303
        # the reference against ITSELF: window 0 decoded alone (the demo's batch of one, :301) against the batched call above
        y1 = G.to_mot(G.decoder(enc_s[:1], enc_c[:1])).numpy()
        out["self_consistency_batch1_vs_all"] = np.float64(np.abs(y1 - out["Y"][:1]).max())
        # AdaIN gains per decoder layer (net/transformer.py:105-113): gamma = style(cha)[..., :dim]
        gam_min, gam_hist = [], []
        x = enc_s
        for nm_, attn, ff in G.decoder.layers:
            style = nm_.style(enc_c.permute(0, 2, 1))
            gamma = style.chunk(2, 1)[0]
            g1 = (1.0 + gamma).abs().numpy().ravel()
            gam_min.append(float(g1.min()))
            gam_hist.append(np.histogram(g1, bins=GAMMA_BINS)[0])
            x = nm_(x, enc_c); x = attn(x, enc_c) + x; x = ff(x) + x
        out["adain_gain_min"] = np.asarray(gam_min)
        out["adain_gain_hist"] = np.asarray(gam_hist)
        out["adain_gain_bins"] = np.asarray(GAMMA_BINS)
    meta = dict(layout=layout, V=V, B=B, seed=a.seed, ckpt_sha256=sha256_of(a.ckpt), ckpt=os.path.basename(a.ckpt),
                norm=bool(a.norm), cnt_norm=bool(a.cnt_norm), torch=torch.__version__, max_abs_Y=float(np.abs(out["Y"]).max()))
    np.savez(a.out, **out, meta=np.array(json.dumps(meta)))
    print(f"wrote {a.out}: {B} + {B} windows, |Y|max {meta['max_abs_Y']:.3g}, reference batch-1 vs batch-{B}: {float(out['self_consistency_batch1_vs_all']):.2e}")
    for l, (m, h) in enumerate(zip(gam_min, gam_hist)):
        print(f"decoder layer {l}: min |1 + gamma| = {m:.3e}; histogram over {GAMMA_BINS[:-1]} -> {h.tolist()}")
    return 0


# ------------------------------------------------------------------------------------------------ GPU half
def replay(a):
    import torch
    from mocha_sigasia2023_amd import ContextBank, Generator, mean_variance_norm
    fx = np.load(a.fixture, allow_pickle=False)
    meta = {}
    if "meta" in fx.files:
        try:
            meta = json.loads(str(fx["meta"]))
        except ValueError:
            meta = eval(str(fx["meta"]), {"__builtins__": {}})            # tests/golden/generator_*.npz: repr of a dict of literals
    layout = meta.get("layout", a.layout)
    if meta.get("ckpt_sha256") and meta["ckpt_sha256"] != sha256_of(a.ckpt):
        print(f"real_asset_parity: {a.ckpt} is not the file the fixture was made from (sha256 differs)", file=sys.stderr)
        return 2
    ck = torch.load(a.ckpt, map_location="cpu")
    state = ck["gen_ema"] if isinstance(ck, dict) and "gen_ema" in ck else ck
    model = Generator(layout=layout, device=a.device).load_state_dict(state).eval()
    dev = torch.device(a.device)
    V = model.V
    (X_mean, X_std, Y_mean, Y_std), (cnt_mean, cnt_std) = load_norms(a.norm, a.cnt_norm, V)
    rows = []

    def stage(name, got, want):
        d = float(np.abs(got.detach().cpu().numpy().astype(np.float64) - want.astype(np.float64)).max())
        rows.append((name, d, float(np.abs(want).max())))

    enc = {}
    for tag in ("src", "cha"):
        X = torch.from_numpy(fx[f"{tag}_X"]).to(dev)
        tokens = model.mot_embedding(X)
        stage(f"{tag}.mot_embedding", tokens, fx[f"{tag}_tokens"])
        encoded = model.encoder(tokens + model.pos_emb[:, :tokens.shape[1]])
        stage(f"{tag}.encoder", encoded, fx[f"{tag}_encoded"])
        cnt = mean_variance_norm(encoded.permute(0, 2, 1)).permute(0, 2, 1)
        stage(f"{tag}.mean_variance_norm", cnt, fx[f"{tag}_cnt"])
        enc[tag] = (encoded, cnt)
    B = fx["src_X"].shape[0]
    idx_ok = None
    if "frame_index" in fx.files:                               # the matching step: the library's exact 1-NN against the reference's BallTree
        cm, cs = torch.from_numpy(cnt_mean).to(dev), torch.from_numpy(cnt_std).to(dev)
        cha_nm = ((enc["cha"][1] - cm) / cs).reshape(B, -1).contiguous()
        src_nm = ((enc["src"][1] - cm) / cs).reshape(B, -1).contiguous()
        bank = ContextBank(model, cha_nm, enc["cha"][0])
        dist, idx = bank.query(src_nm)                          # (Euclidean distance, index) like BallTree.query(k=1)
        idx = idx.cpu().numpy().ravel()
        idx_ok = int((idx == fx["frame_index"]).sum())
        # a differing index is a near-tie unless its distance is further from the reference's than rounding explains
        d_ref = fx["frame_dist"].astype(np.float64)
        d_got = dist.cpu().numpy().ravel().astype(np.float64)
        rows.append(("match.distance", float(np.abs(d_got - d_ref).max()), float(d_ref.max())))
        pick = torch.from_numpy(fx["frame_index"].astype(np.int64)).to(dev)      # decode with the REFERENCE's choice: stages stay comparable
        cha_enc = torch.from_numpy(fx["cha_encoded"]).to(dev)[pick]
    else:
        cha_enc = torch.from_numpy(fx["cha_encoded"]).to(dev)
    src_enc = torch.from_numpy(fx["src_encoded"]).to(dev)
    dec = model.decoder(src_enc, cha_enc)                       # fed with the reference's encoder outputs: the stage's own error
    stage("decoder", dec, fx["decoded"])
    Y = model.to_mot(torch.from_numpy(fx["decoded"]).to(dev))
    stage("to_mot", Y, fx["Y"])
    Yall = model.to_mot(model.decoder(enc["src"][0], enc["cha"][0][pick] if "frame_index" in fx.files else enc["cha"][0]))
    stage("end to end (own encoder outputs) Y", Yall, fx["Y"])
    if "Y_denorm" in fx.files and Y_mean is not None:
        yd = Yall.cpu().numpy() * Y_std[None, None, 1:] + Y_mean[None, None, 1:]
        rows.append(("end to end de-normalised Y", float(np.abs(yd.astype(np.float64) - fx["Y_denorm"]).max()), float(np.abs(fx["Y_denorm"]).max())))
    selfc = float(fx["self_consistency_batch1_vs_all"]) if "self_consistency_batch1_vs_all" in fx.files else None
    print(f"{'stage':42s} {'max |hip - ref|':>16s} {'|ref| max':>11s}   against {TOL:g}")
    worst = 0.0
    for name, d, m in rows:
        print(f"{name:42s} {d:16.3e} {m:11.3e}   {'ok' if d <= TOL else 'ABOVE'}")
        if name.startswith("end to end") or name in ("decoder", "to_mot"):
            worst = max(worst, d)
    if idx_ok is not None:
        print(f"matched indices equal to the reference BallTree's: {idx_ok} of {B}")
    if "adain_gain_min" in fx.files:
        for l, m in enumerate(fx["adain_gain_min"]):
            print(f"decoder layer {l}: min |1 + gamma| = {float(m):.3e}   histogram {fx['adain_gain_hist'][l].tolist()} over edges {fx['adain_gain_bins'][:-1].tolist()}")
    if selfc is not None:
        print(f"the reference against itself (window 0 alone vs in the batch): {selfc:.3e}")
    # the literal tolerance binds wherever the reference reproduces itself to better than it
    bound = TOL if selfc is None else max(TOL, 2.0 * selfc)
    ok = worst <= bound and (idx_ok is None or idx_ok == B or rows[[r[0] for r in rows].index("match.distance")][1] <= 1e-3)
    print(f"RESULT: worst decoder / to_mot / end-to-end difference {worst:.3e} against {bound:.3e}: {'PASS' if ok else 'FAIL'}")
    return 0 if ok else 1


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    sub = ap.add_subparsers(dest="cmd", required=True)
    for name in ("make", "replay"):
        p = sub.add_parser(name)
        p.add_argument("--ckpt", required=True, help="gen_125.pt (the reference Trainer's file: {'gen', 'gen_ema', 'gen_opt'})")
        p.add_argument("--norm", default=None, help="norm.npz (X_mean, X_std, Y_mean, Y_std)")
        p.add_argument("--cnt-norm", default=None, help="cnt_norm.npz (mean, std)")
        p.add_argument("--layout", default="mocha")
    sub.choices["make"].add_argument("--out", required=True)
    sub.choices["make"].add_argument("--windows", type=int, default=16)
    sub.choices["make"].add_argument("--seed", type=int, default=1777)
    sub.choices["replay"].add_argument("--fixture", required=True)
    sub.choices["replay"].add_argument("--device", default="cuda:0")
    a = ap.parse_args(argv)
    return make(a) if a.cmd == "make" else replay(a)


if __name__ == "__main__":
    sys.exit(main())
