"""mocha_sigasia2023_amd — MI355X-native (gfx950) implementation of the MOCHA ``Generator``
hot path (motion encoder -> context matching -> body-part decoder) behind the reference's
own call surface.  The arithmetic lives in ``libmocha_hip.so`` (``csrc/``, C ABI in
``include/mocha_hip.h``); this package is the thin host-side mirror of the reference interface.
"""
from .generator import CVAE, ContextBank, Generator, OursSession, StreamingCharacterizer, mean_variance_norm  # noqa: F401
from .bank import BatchPipeline, ShardedContextBank, build_bank, load_bank, save_bank  # noqa: F401
from .postprocess import PostProcessor, pose_heads, retarget_clip, retarget_clip_ours, write_bvh  # noqa: F401
from .skeleton import skeleton_constants  # noqa: F401
from . import synthetic  # noqa: F401
from .weights import DEFAULT_CFG, param_shapes, synthetic_state_dict  # noqa: F401

__all__ = ["Generator", "CVAE", "OursSession", "ContextBank", "StreamingCharacterizer", "mean_variance_norm", "skeleton_constants",
           "synthetic_state_dict", "param_shapes", "DEFAULT_CFG", "build_bank", "save_bank", "load_bank", "ShardedContextBank", "BatchPipeline",
           "PostProcessor", "pose_heads", "retarget_clip", "retarget_clip_ours", "write_bvh"]
