"""Row N4's tail: the reference's database.bin reader, the step-1 windowing of its bank scripts and the chain database.bin ->
featurise -> encode -> bank files, against a fixture produced by the reference's own reader (etc/utils.py:144-190) and by
executing its collect_CVAE_feature_action.py:93-165 (tests/golden/make_golden.py::run_database)."""
import hashlib
import os

import numpy as np
import pytest
import torch

from mocha_sigasia2023_amd import synthetic, weights
from mocha_sigasia2023_amd.bank import collect_windows, load_database, write_database


@pytest.fixture(scope="module")
def fx(golden_dir):
    return np.load(os.path.join(golden_dir, "database_bank.npz"))


def _db(fx, tmp_path):
    path = str(tmp_path / "database.bin")
    open(path, "wb").write(fx["bin"].tobytes())
    return path, load_database(path)


def test_reader_matches_the_reference_reader(fx, tmp_path):
    path, db = _db(fx, tmp_path)
    for k in fx.files:
        if k.startswith("db_"):
            assert np.array_equal(db[k[3:]], fx[k]), k
        elif k.startswith("dbsha_"):
            shape, dtype, sha = fx[k]
            a = db[k[6:]]
            assert str(a.shape) == shape and str(a.dtype) == dtype, k
            assert hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest() == sha, k
    assert np.array_equal(db["action_labels"], db["content_labels"])
    # writer is the reader's inverse, byte for byte
    out = str(tmp_path / "again.bin")
    write_database(out, db)
    assert open(out, "rb").read() == fx["bin"].tobytes()
    with pytest.raises(ValueError, match="trailing"):
        open(out, "ab").write(b"\0\0\0\0")
        load_database(out)


def test_windowing_matches_the_reference_script(fx, tmp_path):
    _, db = _db(fx, tmp_path)
    w = collect_windows(db, fx["style_keep"], fx["action_keep"])
    assert np.array_equal(w["starts"], fx["starts"])
    assert np.array_equal(w["range_starts"], fx["range_starts"]) and np.array_equal(w["range_stops"], fx["range_stops"])
    assert np.array_equal(w["action_label"], fx["action_label"])
    assert collect_windows(db, [99], [1])["starts"].size == 0        # nothing selected: empty, ragged lists handled
    short = dict(db); short["range_stops"] = db["range_starts"] + 60  # clips of exactly one window length yield no window (range(60, 60))
    assert collect_windows(short, fx["style_keep"], fx["action_keep"])["starts"].size == 0


@pytest.mark.gpu
def test_bank_from_database_on_the_device(fx, tmp_path):
    from mocha_sigasia2023_amd import ContextBank, Generator
    from mocha_sigasia2023_amd.bank import build_bank_from_database, load_bank, save_bank
    from oracle import mocha_oracle as O
    path, db = _db(fx, tmp_path)
    sd = weights.synthetic_state_dict(13, 1.0)
    model = Generator(device="cuda:0").load_state_dict(sd).eval()
    with pytest.raises(RuntimeError, match="set_pose_norm"):
        build_bank_from_database(model, path, fx["style_keep"], fx["action_keep"])
    Ym, Ys = np.zeros((25, 15), np.float32), np.ones((25, 15), np.float32)
    model.set_pose_norm(fx["X_mean"], fx["X_std"], Ym, Ys)
    bank = build_bank_from_database(model, path, fx["style_keep"], fx["action_keep"], batch=7)      # ragged batches
    N = len(fx["starts"])
    assert bank["encoded"].shape == (N, 90, 256)
    # the features the encoder saw = the reference's normalised X (FK + re-rooting + z-score), through the oracle encoder
    with torch.no_grad():
        enc_o, cnt_o = O.encode(O.to_torch_state(sd), torch.from_numpy(fx["X_sel"]))
    sel = fx["sel"]
    assert float((bank["encoded"].cpu()[sel] - enc_o).abs().max()) < 1e-4 * max(1.0, float(enc_o.abs().max()))
    assert float((bank["cnt"].cpu()[sel] - cnt_o).abs().max()) < 1e-4 * max(1.0, float(cnt_o.abs().max()))
    save_bank(str(tmp_path / "feat.npz"), bank, norm_path=str(tmp_path / "cnt_norm.npz"))
    z = load_bank(str(tmp_path / "feat.npz"), str(tmp_path / "cnt_norm.npz"))
    assert np.array_equal(z["range_starts"], fx["range_starts"]) and np.array_equal(z["range_stops"], fx["range_stops"])
    assert np.array_equal(z["action_label"], fx["action_label"])
    assert np.allclose(z["cnt_mean"], bank["cnt"].cpu().numpy().mean(0), atol=2e-5)
    std = np.maximum(z["cnt_std"], 1e-3)
    nm = (z["cnt"] - z["cnt_mean"][None]) / std[None]
    idx = ContextBank(model, torch.from_numpy(nm).cuda(), torch.from_numpy(z["encoded"]).cuda()).query(torch.from_numpy(nm).cuda(), return_distance=False)
    assert idx[:, 0].cpu().tolist() == list(range(N))
