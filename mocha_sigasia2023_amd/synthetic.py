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


def postprocess_inputs(seed: int, N: int, V: int = 24, T: int = 60):
    """Synthetic inputs of the demo's per-frame post-processing (test_fullframework.py:455-540): de-normalised decoded
    windows Y (N,T,V,15) float32 whose last frames form a smooth pose sequence over a fixed random bone-offset
    skeleton, the source's root-local velocities (N,3) x2, its hip velocities over each window (N,T,3) and contact
    labels (N,2) uint8 in runs, so that locks, unlocks and radius-triggered unlocks all occur."""
    r = _rng(seed)
    offs = (0.25 * r.standard_normal((V, 3))).astype(np.float32)
    offs[:, 1] -= 0.15                                   # bones mostly hang downwards
    base = r.standard_normal((V, 4))
    drift = 0.25 * r.standard_normal((V, 4))
    t = np.arange(N, dtype=np.float64)[:, None, None]
    q = base[None] + drift[None] * np.sin(0.07 * t + np.arange(V)[None, :, None])
    q /= np.sqrt((q * q).sum(-1, keepdims=True))
    w, x, y, z = (q[..., i] for i in range(4))
    c0 = np.stack([1 - 2 * (y * y + z * z), 2 * (x * y + w * z), 2 * (x * z - w * y)], -1)      # first / second matrix columns
    c1 = np.stack([2 * (x * y - w * z), 1 - 2 * (x * x + z * z), 2 * (y * z + w * x)], -1)
    xy = np.stack([c0, c1], -1).reshape(N, V, 6)                                                # (..., 3, 2) row-major
    Y = np.empty((N, T, V, 15), np.float32)
    Y[..., 0:3] = offs[None, None] + 0.01 * r.standard_normal((N, T, V, 3))
    Y[..., 3:9] = xy[:, None] + 0.05 * r.standard_normal((N, T, V, 6))
    Y[..., 9:12] = 0.5 * r.standard_normal((N, T, V, 3))
    Y[..., 12:15] = r.standard_normal((N, T, V, 3))
    src_rvel = (np.array([0.1, 0.0, 1.2]) + 0.1 * r.standard_normal((N, 3))).astype(np.float32)
    src_rang = (np.array([0.0, 0.4, 0.0]) + 0.05 * r.standard_normal((N, 3))).astype(np.float32)
    src_hipvel = (0.6 * r.standard_normal((N, T, 3))).astype(np.float32)
    src_hipvel[N // 2] *= 0.05                           # one frame whose speed ratio leaves [0.33, 3] and is reset to 1
    phase = (np.arange(N)[:, None] // 9 + np.array([0, 1])[None]) % 2
    phase[N // 3: N // 3 + 30, 0] = 1                     # a long contact: exercised until the unlock radius trips
    return Y, src_rvel, src_rang, src_hipvel, phase.astype(np.uint8)


def smooth_bone_clip(seed: int, frames: int = 644, J: int = 25, phase: float = 0.0, gain: float = 1.0):
    """A smooth synthetic clip of local bone features, frame by frame - the stand-in for a loaded BVH clip
    (test_fullframework.py:124-139): every bone's rotation swings periodically (a 'gait' of 0.9-2.1 Hz at 60 fps) around its own
    rest orientation, the root travels forward with a vertical bob, bone offsets are FIXED - so the non-root linear velocities
    are exactly zero, constant channels as real clips have them - and the angular velocities are the analytic derivatives'
    size.  `phase` shifts time by a fraction of a frame and `gain` scales the swing: a second clip of the same motion whose
    windows are near-duplicates of the first one's, never equal.  Returns (rot (F,J,4), pos (F,J,3), vel, ang), float32."""
    r = _rng(seed)
    t = (np.arange(frames, dtype=np.float64) + phase)[:, None, None]
    base = r.standard_normal((J, 4)); base /= np.sqrt((base * base).sum(-1, keepdims=True))
    amp = gain * 0.35 * r.uniform(0.2, 1.0, (J, 4))
    om = 2 * np.pi * r.uniform(0.9, 2.1, (J, 1)) / 60.0
    ph = r.uniform(0, 2 * np.pi, (J, 4))
    q = base[None] + amp[None] * np.sin(om[None] * t + ph[None])
    q /= np.sqrt((q * q).sum(-1, keepdims=True))
    q = np.where(q[..., :1] > 0, q, -q)
    offs = 0.25 * r.standard_normal((J, 3)); offs[:, 1] -= 0.15
    pos = np.repeat(offs[None], frames, 0)
    tt = t[:, 0, 0]
    root = np.stack([0.05 * np.sin(0.05 * tt), 0.9 + 0.03 * np.sin(0.21 * tt), 1.2 * tt / 60.0], -1)
    pos[:, 0] = root
    vel = np.zeros((frames, J, 3))
    vel[:, 0] = np.stack([0.05 * 0.05 * 60 * np.cos(0.05 * tt), 0.03 * 0.21 * 60 * np.cos(0.21 * tt), np.full_like(tt, 1.2)], -1)
    ang = gain * 0.35 * 60.0 * om[None] * r.uniform(0.2, 1.0, (J, 3))[None] * np.cos(om[None] * t + ph[None, :, :3])
    return q.astype(np.float32), pos.astype(np.float32), vel.astype(np.float32), ang.astype(np.float32)


def slide_windows(a: np.ndarray, window: int = 60, step: int = 1) -> np.ndarray:
    """(F, ...) -> (F - window + 1, window, ...) windows slid with `step` (process_data(window=60, window_step=1),
    test_fullframework.py:128; preprocess/generate_database.py:65-84), as a contiguous copy."""
    v = np.lib.stride_tricks.sliding_window_view(a, window, axis=0)          # (F - w + 1, ..., w)
    return np.ascontiguousarray(np.moveaxis(v, -1, 1)[::step])
