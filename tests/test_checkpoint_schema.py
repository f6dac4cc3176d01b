"""The checkpoint-file schema, pinned to the reference's own writer (CPU): tests/golden/checkpoint_schema.json is what
Trainer.save_checkpoint (trainer.py:210-222) wrote in the build container and torch.load read back (make_golden.py::
run_checkpoint_schema) - top-level keys, and name / shape / dtype of every 'gen' / 'gen_ema' entry - plus the key list of the CVAE's
bare state_dict (test_fullframework.py:52-58).  weights.py's tables (what load_state_dict expects, what the synthetic weights
are shaped like, what mocha_create's expectations are built from) must be exactly that."""
import json
import os

import numpy as np

from mocha_sigasia2023_amd import weights


def _schema(golden_dir):
    return json.load(open(os.path.join(golden_dir, "checkpoint_schema.json")))


def test_generator_keys_and_shapes_are_the_reference_trainers(golden_dir):
    sc = _schema(golden_dir)
    assert sc["top_level_keys"] == ["gen", "gen_ema", "gen_opt"]                              # trainer.py:219-221
    assert sc["gen_opt_keys"] == ["state", "param_groups"]
    assert sc["gen"] == sc["gen_ema"]                                                          # same module class twice (trainer.py:21-23)
    ours = {k: [list(v), "float32"] for k, v in weights.param_shapes().items()}
    ours.update({k: [list(v.shape), str(v.dtype)] for k, v in weights.buffer_arrays("mocha").items()})
    assert ours == sc["gen_ema"]
    assert sum(int(np.prod(v[0])) for k, v in sc["gen_ema"].items() if k in weights.param_shapes()) == 6_116_559
    sd = weights.synthetic_state_dict(1, 1.0)
    assert {k: [list(v.shape), str(v.dtype)] for k, v in sd.items()} == sc["gen_ema"]
    assert sc["dataparallel_prefix"] == "module." and sc["dataparallel_prefix_checked"] is True   # trainer.py:45-47


def test_cvae_keys_and_shapes_are_the_reference_modules(golden_dir):
    sc = _schema(golden_dir)["cvae"]
    want = {k: [list(v), "float32"] for k, v in weights.cvae_param_shapes().items()}
    used = {k: v for k, v in sc.items() if not k.startswith("encoder.") and not k.endswith("pos_encoder.pe")}
    assert used == want                                                                        # what CVAE.sample touches (model_CVAE.py:44-46)
    rest = set(sc) - set(want)
    assert rest and all(k.startswith("encoder.") or k.endswith("pos_encoder.pe") for k in rest)   # posterior encoder + PE buffers: accepted, unused
    sd = weights.synthetic_cvae_state_dict(99, 1.0)
    assert {k: [list(v.shape), str(v.dtype)] for k, v in sd.items()} == want
    assert sum(int(np.prod(v[0])) for v in sc.values()) > 3_691_008 - 1                         # SURVEY appendix B counts parameters only
