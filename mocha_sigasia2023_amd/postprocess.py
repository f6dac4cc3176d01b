"""Host mirror of the demo's post-processing: decoded windows -> animated skeleton -> BVH.

``pose_heads`` and ``PostProcessor.run`` call the HIP kernels of csrc/postprocess.hip through the C ABI; nothing here
computes on the host except the final text formatting of ``write_bvh`` (the reference's motion/bvh.py:179-224 layout).
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence

import numpy as np
import torch

from .generator import Generator, _dev_f32, _ptr, _stream


class PostCfg(C.Structure):
    """``mocha_post_cfg`` of include/mocha_hip.h; defaults are the demo's constants (test_fullframework.py:104-114)."""
    _fields_ = [("dt", C.c_double), ("ik_max_length_buffer", C.c_double), ("ik_foot_height", C.c_double),
                ("ik_unlock_radius", C.c_double), ("ik_blending_halflife", C.c_double),
                ("ik_enabled", C.c_int), ("n_contact", C.c_int), ("contact_bones", C.c_int * 4), ("blend_enabled", C.c_int)]


def pose_heads(model: Generator, Y):
    """Y (B,60,V,15) de-normalised -> (heads (B,V,13) = [pos | quat wxyz | vel | ang] of the last frame, speed (B,))."""
    Y = _dev_f32(Y, model.device, (model.cfg["nframes"], model.V, model.cfg["mot_in_dim"]), "Y")
    B = Y.shape[0]
    heads = torch.empty((B, model.V, 13), dtype=torch.float32, device=model.device)
    speed = torch.empty((B,), dtype=torch.float32, device=model.device)
    model._ctx.call("mocha_pose_heads", _ptr(Y), B, _ptr(heads), _ptr(speed), _stream())
    return heads, speed


class PostProcessor:
    """Root integration + blending + foot-lock IK of whole clips on the device, one lane per clip."""

    def __init__(self, model: Generator, contact_bones: Optional[Sequence[int]] = None, ik_enabled: bool = True, blend: bool = True, **ik):
        self.model = model
        self.cfg = PostCfg()
        model._ctx.lib.mocha_post_cfg_default(C.byref(self.cfg))
        if contact_bones is not None:
            if len(contact_bones) > 4:
                raise ValueError("at most 4 contact bones")
            self.cfg.n_contact = len(contact_bones)
            for i, b in enumerate(contact_bones):
                self.cfg.contact_bones[i] = int(b)
        self.cfg.ik_enabled = int(ik_enabled)
        self.cfg.blend_enabled = int(blend)     # blend=False, ik_enabled=False: the demo's "cm_" stream (test_fullframework.py:512-527, 637-641)
        for k, v in ik.items():
            if k not in ("dt", "ik_max_length_buffer", "ik_foot_height", "ik_unlock_radius", "ik_blending_halflife"):
                raise TypeError(f"unknown post-processing option {k!r}")
            setattr(self.cfg, k, float(v))

    def run(self, heads, speed, src_rvel, src_rang, src_speed, contact, bvh: bool = True):
        """One clip (N, ...) or a batch of clips (C, N, ...).  Returns a dict of float64 device tensors:
        pos (.., N, V+1, 3), rot / ik_rot (.., N, V+1, 4) and, with ``bvh``, bvh_pos / bvh_euler (.., N, V, 3)."""
        m, dev, V = self.model, self.model.device, self.model.V
        heads = torch.as_tensor(heads)
        single = heads.dim() == 3
        lead = () if single else (heads.shape[0],)

        def f32(a, tail, name):
            return _dev_f32(torch.as_tensor(a).reshape((-1,) + tail), dev, None, name)
        N = heads.shape[-3]
        nclip = 1 if single else heads.shape[0]
        h = f32(heads, (V, 13), "heads")
        sp, ssp = f32(speed, (), "speed"), f32(src_speed, (), "src_speed")
        rv, ra = f32(src_rvel, (3,), "src_rvel"), f32(src_rang, (3,), "src_rang")
        nc = self.cfg.n_contact
        ct = torch.as_tensor(contact).to(device=dev, dtype=torch.uint8).reshape(-1, max(nc, 1)).contiguous()
        rows = nclip * N
        for t, name in ((h, "heads"), (sp, "speed"), (ssp, "src_speed"), (rv, "src_rvel"), (ra, "src_rang"), (ct, "contact")):
            if t.shape[0] != rows:
                raise ValueError(f"postprocess: {name} has {t.shape[0]} rows, expected {rows}")
        out = {"pos": torch.empty(lead + (N, V + 1, 3), dtype=torch.float64, device=dev),
               "rot": torch.empty(lead + (N, V + 1, 4), dtype=torch.float64, device=dev),
               "ik_rot": torch.empty(lead + (N, V + 1, 4), dtype=torch.float64, device=dev)}
        if bvh:
            out["bvh_pos"] = torch.empty(lead + (N, V, 3), dtype=torch.float64, device=dev)
            out["bvh_euler"] = torch.empty(lead + (N, V, 3), dtype=torch.float64, device=dev)
        m._ctx.call("mocha_postprocess", C.byref(self.cfg), _ptr(h), _ptr(sp), _ptr(rv), _ptr(ra), _ptr(ssp), _ptr(ct), nclip, N,
                    _ptr(out["pos"]), _ptr(out["rot"]), _ptr(out["ik_rot"]),
                    _ptr(out["bvh_pos"]) if bvh else None, _ptr(out["bvh_euler"]) if bvh else None, _stream())
        return out


def retarget_clip(bank, src_X, cnt_mean, cnt_std, src_rvel, src_rang, src_speed, contact, raw: bool = False,
                  post: Optional[PostProcessor] = None, bvh: bool = True):
    """The NN branch of the demo from featurised source windows to the animated skeleton, all on the device:
    characterize (encode, match, decode, to_mot; test_fullframework.py:438-443, 465-467) -> pose heads (:457-462) ->
    root integration / blending / foot-lock IK (:492-632) -> BVH channels (:677-681, 697).  ``src_X`` must lead to
    de-normalised windows: pass ``raw=True`` after ``Generator.set_pose_norm`` or de-normalise the result yourself.
    src_speed (N,) is mean_t |src_Yvel[i, t, 1]| (:493)."""
    Y = bank.characterize(src_X, cnt_mean, cnt_std, raw=raw)
    heads, speed = pose_heads(bank.model, Y)
    return (post or PostProcessor(bank.model)).run(heads, speed, src_rvel, src_rang, src_speed, contact, bvh=bvh)


def retarget_clip_ours(session, src_encoded, src_cnt, src_rvel, src_rang, src_speed, contact, eps=None, deterministic: bool = False,
                       post: Optional[PostProcessor] = None, bvh: bool = True, denorm=None):
    """The demo's CVAE ("Ours") branch for a whole clip (test_fullframework.py:446-457 per frame, then :474-632): the
    autoregressive frame loop through ``OursSession`` (already ``reset`` with the first matched character feature), then the
    pose heads and the post-processing of all frames at once.  src_encoded / src_cnt (N,90,256); ``eps`` (N,256) fixes the
    sampler's noise; ``denorm`` = (Y_mean, Y_std) broadcastable to (60,V,15) de-normalises the decoded windows as :457 does."""
    m = session.model
    N = src_encoded.shape[0]
    Ys = []
    for i in range(N):
        y, _ = session.step(src_encoded[i], src_cnt[i], eps=None if eps is None else eps[i:i + 1], deterministic=deterministic)
        Ys.append(y.clone())                    # the session returns views of buffers that the next step overwrites
    Y = torch.cat(Ys)
    if denorm is not None:
        mean, std = (torch.as_tensor(a, dtype=torch.float32, device=m.device) for a in denorm)
        Y = Y * std + mean
    heads, speed = pose_heads(m, Y)
    return (post or PostProcessor(m)).run(heads, speed, src_rvel, src_rang, src_speed, contact, bvh=bvh)


_CHANNEL = {"x": "Xrotation", "y": "Yrotation", "z": "Zrotation"}
_AXIS = {"x": 0, "y": 1, "z": 2}


def write_bvh(path: str, names: Sequence[str], parents: Sequence[int], positions, euler_deg, offsets=None,
              order: str = "zyx", frametime: float = 1.0 / 60.0):
    """BVH text in the reference writer's layout (motion/bvh.py:145-224, save_positions=False): the hierarchy is walked
    depth-first in child-index order, every joint has three rotation channels named after ``order``, the root also three
    position channels, leaves get a zero ``End Site``.  positions / euler_deg (N,V,3); offsets default to the first frame's
    positions (test_fullframework.py:699)."""
    pos = np.asarray(positions.cpu() if torch.is_tensor(positions) else positions, dtype=np.float64)
    rot = np.asarray(euler_deg.cpu() if torch.is_tensor(euler_deg) else euler_deg, dtype=np.float64)
    off = pos[0] if offsets is None else np.asarray(offsets, dtype=np.float64)
    parents = [int(p) for p in parents]
    if not (len(names) == len(parents) == pos.shape[1] == rot.shape[1]):
        raise ValueError("write_bvh: names, parents, positions and rotations disagree on the joint count")
    chan = " ".join(_CHANNEL[c] for c in order)
    ax = [_AXIS[c] for c in order]
    lines, visit = [], [0]

    def joint(i, tabs):
        visit.append(i)
        lines.append(f"{tabs}JOINT {names[i]}"); lines.append(f"{tabs}{{")
        t = tabs + "\t"
        lines.append("%sOFFSET %f %f %f" % (t, off[i, 0], off[i, 1], off[i, 2]))
        lines.append(f"{t}CHANNELS 3 {chan}")
        kids = [j for j, p in enumerate(parents) if p == i]
        for j in kids:
            joint(j, t)
        if not kids:
            lines.append(f"{t}End Site"); lines.append(f"{t}{{")
            lines.append("%s\tOFFSET %f %f %f" % (t, 0.0, 0.0, 0.0))
            lines.append(f"{t}}}")
        lines.append(f"{tabs}}}")

    lines += ["HIERARCHY", f"ROOT {names[0]}", "{"]
    lines.append("\tOFFSET %f %f %f" % (off[0, 0], off[0, 1], off[0, 2]))
    lines.append(f"\tCHANNELS 6 Xposition Yposition Zposition {chan} ")
    for j, p in enumerate(parents):
        if p == 0:
            joint(j, "\t")
    lines += ["}", "MOTION", "Frames: %i" % len(rot), "Frame Time: %f" % frametime]
    # the MOTION block: per frame the root's position, then every visited joint's three angles in channel order, each number as
    # "%f " - one matrix in that column order and ONE format operation for the whole block (a Python loop over frames and joints
    # took 20 ms for a 585-frame clip, several times the GPU's share of the demo)
    n = rot.shape[0]
    if n:
        cols = [pos[:, 0, :]] + [rot[:, j][:, ax] for j in visit]
        M = np.concatenate(cols, axis=1)                            # (n, 3 + 3 * len(visit))
        row_fmt = "%f " * M.shape[1]
        lines.append(("\n".join([row_fmt] * n)) % tuple(M.ravel().tolist()))
    with open(path, "w") as f:
        f.write("\n".join(lines) + "\n")
    return visit
