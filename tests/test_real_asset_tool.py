"""tools/real_asset_parity.py: real-asset parity in one command, validated today on a checkpoint FILE in the
reference Trainer's schema with synthetic weights.

* CPU, build container only (needs /root/reference): the `make` half on a file written by the reference's own
  Trainer.save_checkpoint (trainer.py:210-222) - fixture keys, the BallTree indices, the AdaIN-gain statistics, and the stage outputs
  against the oracle.
* GPU (-m gpu; no reference): the `replay` half loads a checkpoint file through the C ABI host and replays the committed reference
  fixture generator_mocha24_g2.npz (gain 2: outputs of a trained model's size) stage by stage against 1e-4.
"""
import json
import os
import sys

import numpy as np
import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "tools"))
REF = os.environ.get("MOCHA_REFERENCE", "/root/reference")


def _write_trainer_schema_file(path, sd, prefix=""):
    t = {prefix + k: torch.from_numpy(np.array(v)) for k, v in sd.items()}
    p = [torch.nn.Parameter(torch.zeros(3))]
    torch.save({"gen": t, "gen_ema": t, "gen_opt": torch.optim.AdamW(p, lr=1e-4).state_dict()}, path)


@pytest.mark.skipif(not os.path.isdir(REF), reason="the build-container half imports the reference")
@pytest.mark.timeout(900)
def test_make_half_on_a_checkpoint_written_by_the_reference_trainer(tmp_path):
    import real_asset_parity as R
    from mocha_sigasia2023_amd import synthetic, weights
    from oracle import mocha_oracle as O
    sys.path.insert(0, os.path.join(REPO, "tests", "golden"))
    import make_golden as MG
    # the reference's own writer: Trainer(cfg).save_checkpoint with our seeded weights loaded into gen_ema
    cwd = os.getcwd(); os.chdir(REF)
    try:
        for p in (REF, os.path.join(REF, "net"), os.path.join(REF, "etc"), os.path.join(REF, "motion")):
            if p not in sys.path:
                sys.path.append(p)
        from utils import get_config
        from trainer import Trainer
        cfg = get_config(os.path.join(REF, "configs/config.yaml"))
        cfg["model_dir"] = str(tmp_path)
        tr = Trainer(cfg)
        sd = weights.synthetic_state_dict(31, 1.5, "mocha")
        ema = tr.gen_ema.module if hasattr(tr.gen_ema, "module") else tr.gen_ema
        ema.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
        tr.save_checkpoint(125)
    finally:
        os.chdir(cwd)
    ckpt = str(tmp_path / "gen_125.pt")
    rng = np.random.Generator(np.random.PCG64(5))
    norm = str(tmp_path / "norm.npz")
    np.savez(norm, X_mean=rng.standard_normal((25, 15)).astype(np.float32), X_std=(0.5 + rng.random((25, 15))).astype(np.float32),
             Y_mean=rng.standard_normal((25, 15)).astype(np.float32), Y_std=(0.5 + rng.random((25, 15))).astype(np.float32))
    cm, cs = synthetic.cnt_norm(7)
    cnt_norm = str(tmp_path / "cnt_norm.npz")
    np.savez(cnt_norm, mean=cm, std=cs)
    out = str(tmp_path / "real_fixture.npz")
    assert R.main(["make", "--ckpt", ckpt, "--norm", norm, "--cnt-norm", cnt_norm, "--out", out, "--windows", "6", "--seed", "9"]) == 0
    fx = np.load(out)
    meta = json.loads(str(fx["meta"]))
    assert meta["ckpt_sha256"] == R.sha256_of(ckpt) and meta["B"] == 6 and meta["V"] == 24
    for k in ("src_X", "cha_X", "src_tokens", "src_encoded", "src_cnt", "cha_encoded", "src_cnt_nm", "frame_index", "frame_dist",
              "decoded", "Y", "Y_denorm", "adain_gain_min", "adain_gain_hist", "self_consistency_batch1_vs_all"):
        assert k in fx.files, k
    assert fx["adain_gain_hist"].shape == (2, len(R.GAMMA_BINS) - 1) and (fx["adain_gain_hist"].sum(1) == 6 * 256).all()
    assert (fx["adain_gain_min"] > 0).all() and float(fx["self_consistency_batch1_vs_all"]) < 1e-3
    # the stages are the reference's: the oracle (pinned to the reference elsewhere) reproduces them from the same weights and windows
    tsd = O.to_torch_state(sd)
    with torch.no_grad():
        encoded, cnt = O.encode(tsd, torch.from_numpy(fx["src_X"]))
        assert np.abs(encoded.numpy() - fx["src_encoded"]).max() < 1e-4 and np.abs(cnt.numpy() - fx["src_cnt"]).max() < 1e-4
    idx, _ = O.match_bruteforce(fx["src_cnt_nm"].reshape(6, -1), fx["cha_cnt_nm"].reshape(6, -1))
    assert np.array_equal(idx, fx["frame_index"])
    assert np.allclose(fx["Y_denorm"], fx["Y"] * np.load(norm)["Y_std"][None, None, 1:] + np.load(norm)["Y_mean"][None, None, 1:], atol=1e-6)


@pytest.mark.gpu
@pytest.mark.timeout(600)
def test_replay_half_on_a_trainer_schema_file_against_the_reference_fixture(tmp_path, golden_dir, capsys):
    import real_asset_parity as R
    from mocha_sigasia2023_amd import weights
    ckpt = str(tmp_path / "gen_125.pt")
    _write_trainer_schema_file(ckpt, weights.synthetic_state_dict(4242, 2.0, "mocha"), prefix="module.")
    rc = R.main(["replay", "--ckpt", ckpt, "--fixture", os.path.join(golden_dir, "generator_mocha24_g2.npz")])
    text = capsys.readouterr().out
    assert rc == 0, text
    assert "RESULT:" in text and "PASS" in text and "ABOVE" not in text
    for stage in ("src.mot_embedding", "cha.encoder", "decoder", "to_mot", "end to end (own encoder outputs) Y"):
        assert stage in text
    # a checkpoint that is not the fixture's: refused by hash when the fixture records one
    fx = dict(np.load(os.path.join(golden_dir, "generator_mocha24_g2.npz")))
    fx["meta"] = np.array(json.dumps({"layout": "mocha", "ckpt_sha256": "0" * 64}))
    other = str(tmp_path / "other.npz")
    np.savez(other, **fx)
    assert R.main(["replay", "--ckpt", ckpt, "--fixture", other]) == 2
