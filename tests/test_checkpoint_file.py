"""The checkpoint-FILE boundary (run with -m gpu): a `.pt` written with the reference Trainer's schema
(`{'gen', 'gen_ema', 'gen_opt'}` of torch tensors, trainer.py:210-222; schema pinned by tests/golden/checkpoint_schema.json)
loaded with INTEGRATION.md §1's two lines gives bit-identical results to loading the same arrays directly; `module.`-prefixed
DataParallel keys (trainer.py:45-47), strict=False, an extra key, a missing key, and the CVAE's bare state_dict
(test_fullframework.py:47-58)."""
import json
import os

import numpy as np
import pytest
import torch

from mocha_sigasia2023_amd import CVAE, Generator, synthetic, weights

pytestmark = pytest.mark.gpu

# configs/config.yaml:13-44, the `model:` section as get_config returns it (the demo passes cfg['model'] on)
CFG_MODEL = dict(mot_in_dim=15, nframes=60, njoints=24, nbody=6, temporal_patch_size=4, encoder_dim=256, encoder_depth=2, encoder_heads=4,
                 encoder_dim_head=128, encoder_mlp_dim=512, decoder_dim=256, decoder_depth=2, decoder_heads=4, decoder_dim_head=256,
                 decoder_mlp_dim=512, prj_dim=1024, num_patches=-1, num_classes=6,
                 graph=dict(joint=dict(layout="mocha", strategy="distance", max_hop=2), bodypart=dict(layout="mocha", strategy="distance", max_hop=1)))


def _tensors(sd, prefix=""):
    return {prefix + k: torch.from_numpy(np.array(v)) for k, v in sd.items()}


def _write_checkpoint(path, prefix=""):
    """gen and gen_ema carry DIFFERENT weights (as after training: the EMA lags), so loading the wrong one is caught."""
    gen = _tensors(weights.synthetic_state_dict(5, 1.0), prefix)
    ema = _tensors(weights.synthetic_state_dict(6, 1.0), prefix)
    p = [torch.nn.Parameter(torch.zeros(3))]
    opt = torch.optim.AdamW(p, lr=1e-4, weight_decay=1e-4)                                     # trainer.py:36-38
    torch.save({"gen": gen, "gen_ema": ema, "gen_opt": opt.state_dict()}, path)


def _outputs(model):
    src = torch.from_numpy(synthetic.pose_windows(51, 3)).cuda()
    cha = torch.from_numpy(synthetic.pose_windows(52, 3)).cuda()
    tokens = model.mot_embedding(src) + model.pos_emb[:, :90]
    return model(src, cha), model.encoder(tokens)


@pytest.fixture(scope="module")
def reference_outputs():
    m = Generator(device="cuda:0").load_state_dict(weights.synthetic_state_dict(6, 1.0)).eval()          # the ndarray path
    return _outputs(m)


@pytest.mark.parametrize("prefix", ["", "module."])
def test_pt_file_with_the_trainer_schema_loads_through_the_two_integration_lines(tmp_path, golden_dir, reference_outputs, prefix):
    ckpt = str(tmp_path / "gen_125.pt")
    _write_checkpoint(ckpt, prefix)
    schema = json.load(open(os.path.join(golden_dir, "checkpoint_schema.json")))
    on_disk = torch.load(ckpt, map_location="cpu")
    assert list(on_disk.keys()) == schema["top_level_keys"]
    assert {k[len(prefix):]: [list(v.shape), str(v.dtype).replace("torch.", "")] for k, v in on_disk["gen_ema"].items()} == schema["gen_ema"]
    # ---- INTEGRATION.md §1, verbatim
    state = torch.load(ckpt, map_location='cpu')['gen_ema']                   # same file, same keys (trainer.py:239-240)
    model = Generator(CFG_MODEL, device='cuda:0').load_state_dict(state).eval()
    # ----
    Y, enc = _outputs(model)
    assert torch.equal(Y, reference_outputs[0]) and torch.equal(enc, reference_outputs[1])
    assert torch.equal(model.pos_emb.cpu(), state[prefix + "pos_emb"])
    other = Generator(CFG_MODEL, device='cuda:0').load_state_dict(on_disk['gen']).eval()         # 'gen' is another set of weights
    assert not torch.equal(_outputs(other)[0], Y)


def test_strict_and_non_strict_loading(tmp_path, reference_outputs):
    ckpt = str(tmp_path / "gen_125.pt")
    _write_checkpoint(ckpt)
    state = torch.load(ckpt, map_location="cpu")["gen_ema"]
    # an extra key: strict refuses it by name, strict=False ignores it
    extra = dict(state, **{"prj_cnt.mlp_0.0.weight": torch.zeros(4, 4)})                        # e.g. a merged trainer dict (trainer.py:34-35)
    with pytest.raises(KeyError, match="prj_cnt.mlp_0.0.weight"):
        Generator(CFG_MODEL, device="cuda:0").load_state_dict(extra)
    m = Generator(CFG_MODEL, device="cuda:0").load_state_dict(extra, strict=False).eval()
    assert torch.equal(_outputs(m)[0], reference_outputs[0])
    # a missing key: strict names it; strict=False on a FRESH model has no value to fall back on and fails in the library, by name
    short = {k: v for k, v in state.items() if k != "to_mot.6.bias"}
    with pytest.raises(KeyError, match="to_mot.6.bias"):
        Generator(CFG_MODEL, device="cuda:0").load_state_dict(short)
    with pytest.raises(RuntimeError, match="to_mot.6.bias"):
        Generator(CFG_MODEL, device="cuda:0").load_state_dict(short, strict=False)
    # ... on a model that HAS weights, strict=False replaces what the mapping brings and keeps the rest (torch semantics)
    m2 = Generator(CFG_MODEL, device="cuda:0").load_state_dict(torch.load(ckpt, map_location="cpu")["gen"]).eval()
    only_rest = {k: v for k, v in state.items() if k != "to_mot.6.bias"}
    m2.load_state_dict(only_rest, strict=False)
    gen_bias = torch.load(ckpt, map_location="cpu")["gen"]["to_mot.6.bias"]
    mixed = dict(state); mixed["to_mot.6.bias"] = gen_bias
    m3 = Generator(CFG_MODEL, device="cuda:0").load_state_dict(mixed).eval()
    assert torch.equal(_outputs(m2)[0], _outputs(m3)[0])
    # a wrong shape is an error whatever `strict` says
    bad = dict(state); bad["to_mot.6.bias"] = torch.zeros(16)
    with pytest.raises(RuntimeError, match="to_mot.6.bias"):
        Generator(CFG_MODEL, device="cuda:0").load_state_dict(bad, strict=False)
    # the registered graph buffers in the file are cross-checked against the regenerated ones
    wrong = dict(state); wrong["mot_embedding.2.A_j"] = state["mot_embedding.2.A_j"] + 0.01
    with pytest.raises(RuntimeError, match="graph constant"):
        Generator(CFG_MODEL, device="cuda:0").load_state_dict(wrong)
    # half-precision or float64 tensors in a file are converted, not reinterpreted
    f64 = {k: v.double() for k, v in state.items()}
    m4 = Generator(CFG_MODEL, device="cuda:0").load_state_dict(f64).eval()
    assert torch.equal(_outputs(m4)[0], reference_outputs[0])


def test_cvae_bare_state_dict_file(tmp_path, golden_dir):
    """cvae_020000.pt is `torch.save(network_cvae.state_dict())`: every entry of the reference module's state_dict - the
    training-only posterior encoder and the PE buffers included (schema fixture) - loads; results equal the ndarray path."""
    schema = json.load(open(os.path.join(golden_dir, "checkpoint_schema.json")))["cvae"]
    sd = weights.synthetic_cvae_state_dict(99, 1.0)
    r = np.random.Generator(np.random.PCG64(3))
    full = {}
    for k, (shape, dt) in schema.items():
        if k in sd:
            full[k] = torch.from_numpy(sd[k])
        elif k.endswith("pos_encoder.pe"):
            full[k] = torch.from_numpy(np.ascontiguousarray(weights.sincos_pe(shape[1])[None]))
        else:
            full[k] = torch.from_numpy(r.standard_normal(shape).astype(np.float32))            # encoder.*: never read by sample()
    path = str(tmp_path / "cvae_020000.pt")
    torch.save(full, path)
    c = torch.from_numpy(synthetic.token_features(500, 4).reshape(2, 180, 256)).cuda()
    ref = CVAE(device="cuda:0").load_state_dict(sd).eval().sample(c, deterministic=True)
    net = CVAE(device="cuda:0")
    net.load_state_dict(torch.load(path, map_location="cuda:0"))                               # test_fullframework.py:56-58
    assert torch.equal(net.eval().sample(c, deterministic=True), ref)
    pre = CVAE(device="cuda:0").load_state_dict({"module." + k: v for k, v in full.items()}).eval()
    assert torch.equal(pre.sample(c, deterministic=True), ref)
    with pytest.raises(KeyError, match="bogus"):
        CVAE(device="cuda:0").load_state_dict(dict(full, bogus=torch.zeros(1)))
    assert torch.equal(CVAE(device="cuda:0").load_state_dict(dict(full, bogus=torch.zeros(1)), strict=False).eval().sample(c, deterministic=True), ref)
    with pytest.raises(KeyError, match="prior_net.mu_token"):
        CVAE(device="cuda:0").load_state_dict({k: v for k, v in full.items() if k != "prior_net.mu_token"})
