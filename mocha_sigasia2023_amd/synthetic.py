"""Deterministic synthetic inputs shared by the fixture generator, the tests and bench.py.

The reference's data (BVH clips, norm.npz, cnt_norm.npz) is not in its repository
(download.sh:3-25), so every workload is synthetic: z-scored pose windows are N(0, 1)
(the demo z-scores its features, test_fullframework.py:186), banks are N(0, 1)
(SURVEY.md §8d).  numpy's PCG64 stream is stable across numpy versions, so the same seeds
give the same tensors in the build container and on the GPU box.
"""
from __future__ import annotations

import numpy as np


def _rng(seed):
    return np.random.Generator(np.random.PCG64(seed))


def pose_windows(seed: int, B: int, V: int = 24, T: int = 60, C: int = 15) -> np.ndarray:
    """(B, T, V, C) float32, the layout of X at test_fullframework.py:186-189."""
    return _rng(seed).standard_normal((B, T, V, C), dtype=np.float32)


def token_features(seed: int, N: int, ntok: int = 90, dim: int = 256) -> np.ndarray:
    """(N, 90, 256) float32 — stand-in for encoded / cnt bank entries."""
    return _rng(seed).standard_normal((N, ntok, dim), dtype=np.float32)


def temporal_weight(num_temp: int = 15, nbody: int = 6, dim: int = 256) -> np.ndarray:
    """std_weight of the demo: linspace(1, 3, 15) per time patch (train_CVAE.py:64-66),
    broadcast to the (90, 256) token grid (token = t * 6 + v, model.py:49)."""
    w = np.linspace(1.0, 3.0, num_temp, dtype=np.float32)
    return np.repeat(w, nbody)[:, None].repeat(dim, 1).astype(np.float32)


def cnt_norm(seed: int, ntok: int = 90, dim: int = 256):
    """Synthetic cnt_norm.npz stand-in: (mean, std) over the (90, 256) grid with
    ``std`` already divided by the temporal weight (test_fullframework.py:73-76,89)."""
    r = _rng(seed)
    mean = (0.1 * r.standard_normal((ntok, dim))).astype(np.float32)
    std = r.uniform(0.5, 1.5, size=(ntok, dim)).astype(np.float32)
    return mean, (std / temporal_weight(ntok // 6, 6, dim)).astype(np.float32)


def bone_windows(seed: int, B: int, J: int = 25, T: int = 60):
    """Synthetic local bone features of B windows, the inputs of the demo's featurisation
    (test_fullframework.py:135-139): unit quaternions, offsets, linear and angular velocities, float32."""
    r = _rng(seed)
    q = r.standard_normal((B, T, J, 4)).astype(np.float32)
    q /= np.sqrt((q * q).sum(-1, keepdims=True))
    q = np.where(q[..., :1] > 0, q, -q).astype(np.float32)
    pos = (0.3 * r.standard_normal((B, T, J, 3))).astype(np.float32)
    vel = r.standard_normal((B, T, J, 3)).astype(np.float32)
    ang = r.standard_normal((B, T, J, 3)).astype(np.float32)
    return q, pos, vel, ang
