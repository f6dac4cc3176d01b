"""GPU parity of the CVAE sampler against the reference-generated fixture and the oracle."""
import os

import numpy as np
import pytest
import torch

from mocha_sigasia2023_amd import CVAE, synthetic, weights
from oracle import cvae_oracle as CO
from oracle.mocha_oracle import to_torch_state

pytestmark = pytest.mark.gpu


def _model():
    return CVAE(output_seq=90, latent_dim=256, depth=2, nheads=4, feedforward_dim=512, device="cuda:0") \
        .load_state_dict(weights.synthetic_cvae_state_dict(99, 1.0)).eval()


def test_sample_deterministic_matches_reference_fixture(golden_dir):
    z = np.load(os.path.join(golden_dir, "cvae_sample.npz"))
    net = _model()
    c = torch.from_numpy(z["c"]).cuda()
    mu, logvar = net.prior(c)
    out = net.sample(c, deterministic=True)
    assert np.abs(mu.cpu().numpy() - z["mu"]).max() < 1e-4
    assert np.abs(logvar.cpu().numpy() - z["logvar"]).max() < 1e-4
    assert np.abs(out.cpu().numpy() - z["out"]).max() < 1e-4


@pytest.mark.parametrize("B", [1, 5])
def test_sample_with_noise_matches_oracle(B):
    net = _model()
    sd = to_torch_state(weights.synthetic_cvae_state_dict(99, 1.0))
    c = torch.from_numpy(synthetic.token_features(700 + B, 2 * B).reshape(B, 180, 256))
    eps = torch.from_numpy(synthetic.token_features(800 + B, 1)[0, :B].copy())
    with torch.no_grad():
        ref, mu, logvar = CO.sample(sd, c, eps)
    out = net.sample(c.cuda(), eps=eps.cuda())
    assert float((out.cpu() - ref).abs().max()) < 1e-4 * max(1.0, float(ref.abs().max()))
    # autoregressive use (test_fullframework.py:446-452): feeding the output back must stay in tolerance
    cond2 = torch.cat([c[:, :90], ref], dim=1)
    with torch.no_grad():
        ref2, _, _ = CO.sample(sd, cond2)
    out2 = net.sample(torch.cat([c[:, :90].cuda(), out], dim=1), deterministic=True)
    assert float((out2.cpu() - ref2).abs().max()) < 2e-4 * max(1.0, float(ref2.abs().max()))


def test_cvae_errors():
    with pytest.raises(KeyError):
        CVAE(device="cuda:0").load_state_dict({"prior_net.mu_token": np.zeros((1, 1, 256), np.float32)})
    with pytest.raises(RuntimeError, match="not loaded"):
        CVAE(device="cuda:0").sample(torch.zeros(1, 180, 256))


def test_ours_branch_frames_match_oracle():
    """The demo's CVAE branch for a few frames (test_fullframework.py:446-457) against the same loop on the oracles."""
    from mocha_sigasia2023_amd import Generator, OursSession
    from oracle import mocha_oracle as O
    gsd = weights.synthetic_state_dict(12, 1.2)
    csd = weights.synthetic_cvae_state_dict(99, 1.0)
    model = Generator(device="cuda:0").load_state_dict(gsd).eval()
    net = _model()
    rng = np.random.Generator(np.random.PCG64(3))
    stats = [(0.1 * rng.standard_normal((90, 256))).astype(np.float32), rng.uniform(0.5, 1.5, (90, 256)).astype(np.float32),
             (0.1 * rng.standard_normal((90, 256))).astype(np.float32), rng.uniform(0.5, 1.5, (90, 256)).astype(np.float32)]
    src = torch.from_numpy(synthetic.pose_windows(40, 4))
    cha = torch.from_numpy(synthetic.pose_windows(41, 1))
    tg, tc = O.to_torch_state(gsd), to_torch_state(csd)
    with torch.no_grad():
        enc_s, cnt_s = O.encode(tg, src)
        enc_c, _ = O.encode(tg, cha)
    sess = OursSession(model, net, *stats).reset(enc_c[0].cuda())
    prev = enc_c[:1].clone()
    sm, ss, cm, cs = (torch.from_numpy(a) for a in stats)
    for i in range(1, 4):
        eps = torch.from_numpy(synthetic.token_features(900 + i, 1)[0, :1].copy())
        with torch.no_grad():
            cond = torch.cat([(cnt_s[i:i + 1] - sm) / ss, (prev - cm) / cs], dim=1)
            vae, _, _ = CO.sample(tc, cond, eps)
            cur = vae * cs + cm
            Yo = O.to_mot(tg, O.decoder(tg, enc_s[i:i + 1], cur))
            prev = cur
        Y, gcur = sess.step(enc_s[i].cuda(), cnt_s[i].cuda(), eps=eps.cuda())
        assert float((gcur.cpu() - cur).abs().max()) < 1e-4 * max(1.0, float(cur.abs().max())) * i
        assert float((Y.cpu() - Yo).abs().max()) < 1e-4 * max(1.0, float(Yo.abs().max())) * i


def test_ours_session_advances_several_clips_in_lock_step():
    """B clips through one OursSession (clips are the data-parallel axis of the autoregressive CVAE branch) give the same
    frames as B single-clip sessions (the same kernels at another batch size: fp32 summation order may differ)."""
    from mocha_sigasia2023_amd import Generator, OursSession
    gsd = weights.synthetic_state_dict(12, 1.2)
    model = Generator(device="cuda:0").load_state_dict(gsd).eval()
    net = _model()
    rng = np.random.Generator(np.random.PCG64(4))
    stats = [(0.1 * rng.standard_normal((90, 256))).astype(np.float32), rng.uniform(0.5, 1.5, (90, 256)).astype(np.float32),
             (0.1 * rng.standard_normal((90, 256))).astype(np.float32), rng.uniform(0.5, 1.5, (90, 256)).astype(np.float32)]
    B, F = 3, 3
    X = torch.from_numpy(synthetic.pose_windows(50, B * (F + 1))).cuda()
    enc, cnt = model.encode(X)
    enc, cnt = enc.reshape(B, F + 1, 90, 256), cnt.reshape(B, F + 1, 90, 256)
    eps = torch.from_numpy(synthetic.token_features(77, F)[:, :B].copy()).cuda()          # (F, B, 256)
    multi = OursSession(model, net, *stats).reset(enc[:, 0].contiguous())
    singles = [OursSession(model, net, *stats).reset(enc[b, 0]) for b in range(B)]
    for f in range(1, F + 1):
        Ym, cm_ = multi.step(enc[:, f].contiguous(), cnt[:, f].contiguous(), eps=eps[f - 1].contiguous())
        assert Ym.shape == (B, 60, 24, 15)
        for b in range(B):
            Ys, cs_ = singles[b].step(enc[b, f], cnt[b, f], eps=eps[f - 1, b:b + 1].contiguous())
            assert float((Ym[b] - Ys[0]).abs().max()) < 2e-5 * max(1.0, float(Ys.abs().max()))
            assert float((cm_[b] - cs_[0]).abs().max()) < 2e-5 * max(1.0, float(cs_.abs().max()))
    with pytest.raises(ValueError):
        multi.step(enc[0, 1], cnt[0, 1])
