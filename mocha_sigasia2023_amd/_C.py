"""ctypes binding of libmocha_hip.so (include/mocha_hip.h).

There is deliberately no fallback: if the HIP library is missing or a call fails, a
RuntimeError is raised.  The product path never imports the CPU oracle.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmocha_hip.so")

ABI_VERSION = 5


class mocha_cfg(C.Structure):
    _fields_ = [(n, C.c_int) for n in (
        "T", "V", "C_in", "patch", "dim",
        "enc_depth", "enc_heads", "enc_dim_head", "enc_mlp",
        "dec_depth", "dec_heads", "dec_dim_head", "dec_mlp", "layout")]


class mocha_comm_info_t(C.Structure):
    _fields_ = [("nranks", C.c_int), ("rank", C.c_int), ("rccl_version", C.c_int), ("device", C.c_int),
                ("pci_bus_id", C.c_char * 32), ("library", C.c_char * 512)]


_vp, _i, _i64 = C.c_void_p, C.c_int, C.c_int64

# name -> (restype, argtypes); every symbol include/mocha_hip.h declares
SIGNATURES = {
    "mocha_abi_version": (_i, []),
    "mocha_create": (_i, [C.POINTER(mocha_cfg), _i, C.POINTER(_vp)]),
    "mocha_destroy": (None, [_vp]),
    "mocha_last_error": (C.c_char_p, [_vp]),
    "mocha_load_weight": (_i, [_vp, C.c_char_p, _vp, C.POINTER(_i64), _i]),
    "mocha_finalize_weights": (_i, [_vp]),
    "mocha_reserve": (_i, [_vp, _i]),
    "mocha_pos_emb": (_i, [_vp, C.POINTER(_vp)]),
    "mocha_embed": (_i, [_vp, _vp, _i, _vp, _i, _vp]),
    "mocha_encoder": (_i, [_vp, _vp, _i, _vp, _vp]),
    "mocha_mvn": (_i, [_vp, _vp, _i, _vp, _vp, _vp, _vp, _vp]),
    "mocha_encode": (_i, [_vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "mocha_decoder": (_i, [_vp, _vp, _vp, _i, _vp, _vp]),
    "mocha_to_mot": (_i, [_vp, _vp, _i, _vp, _vp]),
    "mocha_style_constants": (_i, [_vp, _vp, _i, _vp, _vp]),
    "mocha_forward": (_i, [_vp, _vp, _vp, _i, _vp, _vp]),
    "mocha_forward_features": (_i, [_vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp]),
    "mocha_bank_set": (_i, [_vp, _vp, _vp, _i64, _i, _vp]),
    "mocha_match": (_i, [_vp, _vp, _i, _vp, _vp, _vp]),
    "mocha_bank_gather": (_i, [_vp, _vp, _i, _vp, _vp]),
    "mocha_match_topk": (_i, [_vp, _vp, _i, _i, _vp, _vp, _vp]),
    "mocha_bank_gather_blend": (_i, [_vp, _vp, _vp, C.c_float, _i, _i, _vp, _vp]),
    "mocha_characterize": (_i, [_vp, _vp, _i, _vp, _vp, _vp, _vp, _vp]),
    "mocha_set_pose_norm": (_i, [_vp, _vp, _vp, _vp, _vp]),
    "mocha_encode_raw": (_i, [_vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "mocha_characterize_raw": (_i, [_vp, _vp, _i, _vp, _vp, _vp, _vp, _vp]),
    "mocha_characterize_pair": (_i, [_vp, _vp, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "mocha_characterize_pair_raw": (_i, [_vp, _vp, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "mocha_cvae_load_weight": (_i, [_vp, C.c_char_p, _vp, C.POINTER(_i64), _i]),
    "mocha_cvae_finalize": (_i, [_vp]),
    "mocha_cvae_sample": (_i, [_vp, _vp, _i, _vp, _vp, _vp, _vp, _vp]),
    "mocha_cvae_condition": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp]),
    "mocha_scale_shift": (_i, [_vp, _vp, _vp, _vp, _i, _vp, _vp]),
    "mocha_featurize": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _vp, _vp]),
    "mocha_post_cfg_default": (None, [_vp]),
    "mocha_pose_heads": (_i, [_vp, _vp, _i, _vp, _vp, _vp]),
    "mocha_postprocess": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "mocha_column_stats": (_i, [_vp, _vp, _i64, _vp, _vp, _vp]),
    "mocha_set_option": (_i, [_vp, C.c_char_p, _i]),
    "mocha_linear": (_i, [_vp, _vp, _vp, _vp, _vp, _i64, _i, _i, _i, _vp]),
    "mocha_graph_constants": (_i, [_vp, _vp, _vp, _vp, _vp]),
    "mocha_generation": (_i64, [_vp]),
    "mocha_step_graph": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _vp]),
    "mocha_step_graph_lane": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _vp]),
    "mocha_set_rccl_library": (_i, [C.c_char_p]),
    "mocha_comm_unique_id": (_i, [_vp, _vp]),
    "mocha_comm_init": (_i, [_vp, _vp, _i, _i]),
    "mocha_comm_destroy": (_i, [_vp]),
    "mocha_comm_info": (_i, [_vp, C.POINTER(mocha_comm_info_t)]),
    "mocha_bank_broadcast": (_i, [_vp, _vp, _i, _i64, _i, _vp]),
    "mocha_bcast_plan": (_i, [_i64, _i, _i, C.POINTER(_i64)]),
    "mocha_build_info": (C.c_char_p, []),
    "mocha_runtime_version": (_i, []),
    "mocha_bank_export": (_i, [_vp, _vp, _vp, _vp]),
    "mocha_scan_byte_state": (_i, [_vp, _i, C.POINTER(C.c_int32), _vp]),
    "mocha_bank_view": (_i, [_vp] + [C.POINTER(_vp)] * 5 + [C.POINTER(_i64)]),
    "mocha_profile_start": (_i, [_vp]),
    "mocha_profile_stop": (_i, [_vp, C.c_char_p, _i64]),
}

_lib = None


def load_library(path: str = LIB_PATH):
    """dlopen the library and bind every declared symbol; raises RuntimeError when absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(path):
        raise RuntimeError(
            f"libmocha_hip.so not found at {path}: build it with "
            "`python -c 'import __graft_entry__ as g; g.build()'` or `make -C mocha_sigasia2023_amd/csrc`. "
            "There is no CPU fallback.")
    try:
        lib = C.CDLL(path)
    except OSError as e:  # missing ROCm runtime etc.
        raise RuntimeError(f"cannot load {path}: {e}") from e
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)      # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    if lib.mocha_abi_version() != ABI_VERSION:
        raise RuntimeError(f"libmocha_hip.so ABI {lib.mocha_abi_version()} != binding ABI {ABI_VERSION}")
    _lib = lib
    return lib


def check(lib, ctx, rc: int, what: str):
    if rc != 0:
        msg = lib.mocha_last_error(ctx)
        raise RuntimeError(f"{what} failed (status {rc}): {msg.decode() if msg else '?'}")
