"""Pins the CVAE oracle against the fixture produced by the reference CVAE (tests/golden/make_golden.py)."""
import os

import numpy as np
import torch

from mocha_sigasia2023_amd import weights
from oracle import cvae_oracle as CO
from oracle.mocha_oracle import to_torch_state


def test_cvae_oracle_matches_reference(golden_dir):
    z = np.load(os.path.join(golden_dir, "cvae_sample.npz"))
    sd = to_torch_state(weights.synthetic_cvae_state_dict(99, 1.0))
    with torch.no_grad():
        out, mu, logvar = CO.sample(sd, torch.from_numpy(z["c"]))
    assert np.abs(mu.numpy() - z["mu"]).max() < 5e-6
    assert np.abs(logvar.numpy() - z["logvar"]).max() < 5e-6
    assert np.abs(out.numpy() - z["out"]).max() < 5e-6


def test_cvae_schema():
    n = sum(int(np.prod(s)) for s in weights.cvae_param_shapes().values())
    assert n == 3691008 - 1054720         # all CVAE parameters minus the training-only posterior encoder (= size of prior_net)
