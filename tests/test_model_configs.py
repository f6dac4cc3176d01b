"""Model configurations other than the shipped one (run with -m gpu).  The reference builds its Generator from configs/config.yaml
(model.py:16-80): depths, head counts, head dims and feed-forward widths are free there; mocha_create accepts depth 1..8, dim_head
64 / 128 / 256 on either side, heads x dim_head <= 1024 and mlp_dim a multiple of 64 up to 2048 (everything else fails loudly,
tests/test_cabi.py; INTEGRATION.md lists what is rejected and why).  Every accepted variation must still be the
reference's arithmetic: compared with the oracle on seeded weights of those shapes, through forward, the demo's characterize sequence,
and at batch sizes that select the small-batch kernels (skinny GEMMs, twelve-wave attention for head dim 256 on EITHER side) and the
tiled / plane engines."""
import numpy as np
import pytest
import torch

from mocha_sigasia2023_amd import ContextBank, Generator, synthetic, weights
from oracle import mocha_oracle as O

pytestmark = pytest.mark.gpu
TOL = 1e-4

CONFIGS = {
    "enc 1 x (4 x 128), dec 3 x (4 x 256)": dict(encoder_depth=1, decoder_depth=3),
    "enc 2 x (4 x 256), dec 2 x (4 x 128)": dict(encoder_dim_head=256, decoder_dim_head=128),
    "enc 3 x (8 x 128), dec 1 x (2 x 256)": dict(encoder_depth=3, encoder_heads=8, decoder_depth=1, decoder_heads=2),
    "enc 2 x (1 x 128), dec 4 x (3 x 256)": dict(encoder_heads=1, decoder_depth=4, decoder_heads=3),
    "enc 8 x (2 x 256), dec 8 x (8 x 128)": dict(encoder_depth=8, encoder_heads=2, encoder_dim_head=256, decoder_depth=8, decoder_heads=8,
                                                 decoder_dim_head=128),
    # the reference Attention's default head dim 64 (net/transformer.py:38) and feed-forward widths other than 512
    "enc 2 x (4 x 64) mlp 256, dec 2 x (8 x 64) mlp 1024": dict(encoder_dim_head=64, decoder_heads=8, decoder_dim_head=64, encoder_mlp_dim=256,
                                                                decoder_mlp_dim=1024),
    "enc 2 x (16 x 64) mlp 2048, dec 2 x (4 x 256) mlp 64": dict(encoder_heads=16, encoder_dim_head=64, encoder_mlp_dim=2048, decoder_mlp_dim=64),
    "enc 2 x (4 x 128) mlp 768, dec 2 x (4 x 256) mlp 320": dict(encoder_mlp_dim=768, decoder_mlp_dim=320),
}


@pytest.mark.parametrize("name", list(CONFIGS))
@pytest.mark.parametrize("B", [2, 40])
def test_other_model_configurations_against_the_oracle(name, B):
    cfg = dict(weights.DEFAULT_CFG, **CONFIGS[name])
    layout = "mocha"
    sd = weights.synthetic_state_dict(4242, 1.0 if cfg["encoder_depth"] + cfg["decoder_depth"] > 8 else 1.3, layout, cfg=cfg)
    model = Generator(cfg, layout=layout, device="cuda:0").load_state_dict(sd).eval()
    src = synthetic.pose_windows(11, B, 24)
    cha = synthetic.pose_windows(12, B, 24)
    mean, std = synthetic.cnt_norm(5)
    ost = O.to_torch_state(sd)
    with O.heads_config(cfg["encoder_heads"], cfg["decoder_heads"]), torch.no_grad():
        ref_fwd = O.generator_forward(ost, torch.from_numpy(src), torch.from_numpy(cha)).numpy()
        ref_enc, ref_cnt = O.encode(ost, torch.from_numpy(cha))
        ref_Y, ref_idx = O.characterize(ost, torch.from_numpy(src), torch.from_numpy(cha), mean, std)
    ts, tc = torch.from_numpy(src).cuda(), torch.from_numpy(cha).cuda()
    tm, tsd = torch.from_numpy(mean).cuda(), torch.from_numpy(std).cuda()
    with torch.no_grad():
        Y = model(ts, tc)
        enc, cnt, nm = model.encode(tc, tm, tsd)
        Yc, idx = ContextBank(model, nm.reshape(B, -1), enc).characterize(ts, tm, tsd, return_index=True)
        Yp, idxp = model.characterize_pair(ts, tc, tm, tsd, return_index=True)
    scale = max(1.0, float(np.abs(ref_fwd).max()))
    assert np.abs(Y.cpu().numpy() - ref_fwd).max() < TOL * scale, name
    es = max(1.0, float(ref_enc.abs().max()))
    assert float((enc.cpu() - ref_enc).abs().max()) < TOL * es and float((cnt.cpu() - ref_cnt).abs().max()) < TOL * max(1.0, float(ref_cnt.abs().max()))
    same = idx.view(-1).cpu().numpy() == ref_idx
    assert same.mean() >= 0.9, (name, same.mean())               # near-ties of the synthetic features may legitimately swap
    ys = max(1.0, float(ref_Y.abs().max()))
    assert float((Yc.cpu()[same] - ref_Y[same]).abs().max()) < TOL * ys
    assert torch.equal(idxp.view(-1), idx.view(-1)) and float((Yp - Yc).abs().max()) < 1e-5 * ys


def test_other_model_configurations_against_the_reference_fixture():
    """The same through fixtures the reference itself produced for four such configurations (tests/golden/make_golden.py::
    run_config_variants): the HIP path, not only the oracle, is held to the reference's outputs."""
    import ast
    import os
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "generator_config_variants.npz"))
    meta = ast.literal_eval(str(z["meta"]))
    src = torch.from_numpy(synthetic.pose_windows(meta["src_seed"], meta["B"], 24)).cuda()
    cha = torch.from_numpy(synthetic.pose_windows(meta["cha_seed"], meta["B"], 24)).cuda()
    for name, ov in meta["variants"].items():
        cfg = dict(weights.DEFAULT_CFG, **ov)
        sd = weights.synthetic_state_dict(meta["seed"], meta["gain"], "mocha", cfg=cfg)
        model = Generator(cfg, layout="mocha", device="cuda:0").load_state_dict(sd).eval()
        with torch.no_grad():
            enc, _ = model.encode(cha)
            Y = model(src, cha)
        assert np.abs(enc.cpu().numpy() - z[f"{name}_cha_encoded"]).max() < TOL * max(1.0, float(np.abs(z[f"{name}_cha_encoded"]).max())), name
        assert np.abs(Y.cpu().numpy() - z[f"{name}_Y_forward"]).max() < TOL, name


@pytest.mark.parametrize("override,message", [
    (dict(encoder_dim_head=96), "dim_head must be 64, 128 or 256"),
    (dict(decoder_dim_head=32), "dim_head must be 64, 128 or 256"),
    (dict(encoder_heads=16, encoder_dim_head=128), "heads\\*dim_head"),
    (dict(encoder_mlp_dim=100), "mlp_dim must be a multiple of 64"),
    (dict(decoder_mlp_dim=4096), "mlp_dim must be a multiple of 64"),
    (dict(encoder_depth=9), "depth must be in"),
    (dict(nframes=64), "unsupported T/patch/dim/C_in"),
    (dict(temporal_patch_size=2), "unsupported T/patch/dim/C_in"),
    (dict(encoder_dim=512, decoder_dim=512), "unsupported T/patch/dim/C_in"),
    (dict(mot_in_dim=12), "unsupported T/patch/dim/C_in"),
])
def test_rejected_configurations_fail_loudly_in_mocha_create(override, message):
    """What the library does NOT accept is refused by mocha_create with a message that names the dimension ('Accepted
    configurations'): the kernels are specialised for T = 60, patch 4, dim 256, 15 input channels (configs/config.yaml:13-31)."""
    with pytest.raises(RuntimeError, match=message):
        Generator(dict(weights.DEFAULT_CFG, **override), layout="mocha", device="cuda:0")
