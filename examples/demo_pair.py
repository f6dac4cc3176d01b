#!/usr/bin/env python3
"""The reference demo (test_fullframework.py) end to end on synthetic data, entirely through the HIP path.

    local bone features of two clips  --featurize-->  X_raw  --(z-score fused)-->  encoder  -->  bank from the character clip
    NN branch   : characterize_pair -> pose heads -> root integration / foot-lock IK -> BVH        ("cm_" stream of the demo)
    Ours branch : CVAE session frame loop -> decoder -> pose heads -> post-processing -> BVH        (the demo's main output)

The reference's data (BVH clips, norm.npz, checkpoints) is not redistributable, so weights, norms and motions are synthetic;
with real assets replace `synthetic_state_dict` by torch.load(ckpt)['gen_ema'], the norms by norm.npz / cnt_norm.npz /
cvae_norm.npz and the bone features by the output of the reference's process_data (motion/bvh.py + preprocess).
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mocha_sigasia2023_amd import (CVAE, ContextBank, Generator, OursSession, retarget_clip, retarget_clip_ours, synthetic,  # noqa: E402
                                   synthetic_state_dict, write_bvh)
from mocha_sigasia2023_amd import weights as W  # noqa: E402
from mocha_sigasia2023_amd.skeleton import LAYOUTS  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=120, help="windows per clip (the demo clips have 585)")
ap.add_argument("--out", default="gpurun_out/demo")
a = ap.parse_args()
os.makedirs(a.out, exist_ok=True)
dev = torch.device("cuda:0")
N = a.frames

# ---- model, CVAE and (synthetic) statistics                                                  test_fullframework.py:40-98
model = Generator(device=dev).load_state_dict(synthetic_state_dict(1777, 1.0)).eval()
cvae = CVAE(device=dev).load_state_dict(W.synthetic_cvae_state_dict(99, 1.0)).eval()
rng = np.random.Generator(np.random.PCG64(0))
J, C = 25, 15
X_mean = (0.05 * rng.standard_normal((J, C))).astype(np.float32); X_std = rng.uniform(0.5, 1.5, (J, C)).astype(np.float32)
Y_mean = (0.05 * rng.standard_normal((J, C))).astype(np.float32); Y_std = rng.uniform(0.2, 0.6, (J, C)).astype(np.float32)
model.set_pose_norm(X_mean, X_std, Y_mean, Y_std)
cnt_mean, cnt_std = synthetic.cnt_norm(7)
stats = [(0.1 * rng.standard_normal((90, 256))).astype(np.float32), rng.uniform(0.5, 1.5, (90, 256)).astype(np.float32),
         (0.1 * rng.standard_normal((90, 256))).astype(np.float32), rng.uniform(0.5, 1.5, (90, 256)).astype(np.float32)]

# ---- the two clips as local bone features of N sixty-frame windows each                      :120-140, 203-222
src_bones = synthetic.bone_windows(11, N)
cha_bones = synthetic.bone_windows(12, N)
_, rvel, rang, hipvel, contact = synthetic.postprocess_inputs(5, N)           # root-local velocities / contacts of the source
src_speed = np.linalg.norm(hipvel, axis=-1).mean(-1).astype(np.float32)

torch.cuda.synchronize(); t0 = time.perf_counter()
src_X = model.featurize(*src_bones)                                            # :141-185 on the device
cha_X = model.featurize(*cha_bones)
# ---- NN branch for the whole clip                                                            :188-194, 271-302, 438-443, 465-467
Y, idx, cha_enc, cha_nm = model.characterize_pair(src_X, cha_X, cnt_mean, cnt_std, return_index=True, return_bank=True, raw=True)
bank = ContextBank(model, cha_nm, cha_enc)
# the reference's "cm_" stream takes the decoded poses as they are: no blending with the previous frame, no foot-lock IK
# (test_fullframework.py:512-527, 637-641)
from mocha_sigasia2023_amd import PostProcessor  # noqa: E402
nn = retarget_clip(bank, src_X, cnt_mean, cnt_std, rvel, rang, src_speed, contact, raw=True,
                   post=PostProcessor(model, ik_enabled=False, blend=False))
# ---- Ours branch: CVAE frame loop seeded with the first matched character feature            :298, 436, 446-457
src_enc, src_cnt = model.encode(src_X, raw=True)
sess = OursSession(model, cvae, *stats).reset(cha_enc[int(idx[0])])
ours = retarget_clip_ours(sess, src_enc[1:], src_cnt[1:], rvel[1:], rang[1:], src_speed[1:], contact[1:],
                          denorm=(Y_mean[1:], Y_std[1:]))
torch.cuda.synchronize(); dt = time.perf_counter() - t0

names = ["Joint%02d" % i for i in range(24)]
parents = LAYOUTS["mocha"]["parents"]
write_bvh(os.path.join(a.out, "cm_trans.bvh"), names, parents, nn["bvh_pos"], nn["bvh_euler"])          # :690-713
write_bvh(os.path.join(a.out, "ours.bvh"), names, parents, ours["bvh_pos"], ours["bvh_euler"])
print(f"{N} windows per clip: featurize + NN branch + Ours branch + post-processing in {dt * 1e3:.1f} ms "
      f"({(N - 1) / dt:.0f} frames/s including the sequential CVAE loop); matched entries {idx[:8].tolist()} ...")
for k in ("cm_trans.bvh", "ours.bvh"):
    p = os.path.join(a.out, k)
    print(f"  {p}: {os.path.getsize(p)} bytes, {sum(1 for _ in open(p))} lines")
assert all(torch.isfinite(v).all() for v in nn.values()) and all(torch.isfinite(v).all() for v in ours.values())
