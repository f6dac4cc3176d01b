"""Row N2: window featurisation.  CPU: oracle vs the fixture made with the reference's quaternion library.
GPU: HIP kernel vs fixture, and the featurise -> encode(raw) chain vs the oracles."""
import os

import numpy as np
import pytest
import torch

from mocha_sigasia2023_amd import synthetic, weights
from mocha_sigasia2023_amd.skeleton import LAYOUTS
from oracle import featurize_oracle as FO


def test_featurize_oracle_matches_reference(golden_dir):
    z = np.load(os.path.join(golden_dir, "featurize.npz"))
    seed, B = (int(v) for v in z["seed"])
    X = FO.featurize(*synthetic.bone_windows(seed, B), FO.full_parents(LAYOUTS["mocha"]["parents"]))
    assert np.array_equal(X, z["X"])


@pytest.mark.gpu
def test_featurize_hip_matches_reference(golden_dir):
    from mocha_sigasia2023_amd import Generator
    z = np.load(os.path.join(golden_dir, "featurize.npz"))
    seed, B = (int(v) for v in z["seed"])
    model = Generator(device="cuda:0")
    X = model.featurize(*(torch.from_numpy(a) for a in synthetic.bone_windows(seed, B)))
    assert X.shape == z["X"].shape
    assert float(np.abs(X.cpu().numpy() - z["X"]).max()) < 1e-4 * max(1.0, float(np.abs(z["X"]).max()))


@pytest.mark.gpu
def test_featurize_to_encode_chain():
    from mocha_sigasia2023_amd import Generator
    from oracle import mocha_oracle as O
    sd = weights.synthetic_state_dict(8, 1.0)
    model = Generator(device="cuda:0").load_state_dict(sd).eval()
    rng = np.random.Generator(np.random.PCG64(1))
    Xm = rng.standard_normal((25, 15)).astype(np.float32); Xs = rng.uniform(1.0, 3.0, (25, 15)).astype(np.float32)
    model.set_pose_norm(Xm, Xs, np.zeros((25, 15), np.float32), np.ones((25, 15), np.float32))
    bones = synthetic.bone_windows(77, 5)
    enc, cnt = model.encode(model.featurize(*(torch.from_numpy(a) for a in bones)), raw=True)
    Xo = FO.featurize(*bones, FO.full_parents(LAYOUTS["mocha"]["parents"]))
    Xn = (Xo[:, :, 1:] - Xm[None, None, 1:]) / Xs[None, None, 1:]           # test_fullframework.py:186
    with torch.no_grad():
        eo, co = O.encode(O.to_torch_state(sd), torch.from_numpy(Xn))
    assert float((enc.cpu() - eo).abs().max()) < 1e-4 * max(1.0, float(eo.abs().max()))
