"""CPU ORACLE for the demo's window featurisation — TEST INFRASTRUCTURE.

NumPy restatement of test_fullframework.py:141-185 (the X features; the Y features of :161-176 are
not inputs of the network): forward kinematics with velocities (motion/quat.py:189-204), re-rooting of
every window on its own last frame (:148-151) and expression of all bones in that root frame
(:154-158), concatenated as [pos 3 | rot-matrix xy 6 | vel 3 | ang 3] (:180-185).  Pinned against the
reference's own quaternion library on synthetic inputs (tests/golden/featurize.npz)."""
from __future__ import annotations

import numpy as np


def q_mul(x, y):                      # motion/quat.py:112-120
    x0, x1, x2, x3 = (x[..., i:i + 1] for i in range(4))
    y0, y1, y2, y3 = (y[..., i:i + 1] for i in range(4))
    return np.concatenate([y0 * x0 - y1 * x1 - y2 * x2 - y3 * x3,
                           y0 * x1 + y1 * x0 - y2 * x3 + y3 * x2,
                           y0 * x2 + y1 * x3 + y2 * x0 - y3 * x1,
                           y0 * x3 - y1 * x2 + y2 * x1 + y3 * x0], axis=-1)


def q_inv(q):                         # :109-110
    return np.asarray([1, -1, -1, -1], dtype=np.float32) * q


def cross(a, b):                      # :3-7
    return np.concatenate([a[..., 1:2] * b[..., 2:3] - a[..., 2:3] * b[..., 1:2],
                           a[..., 2:3] * b[..., 0:1] - a[..., 0:1] * b[..., 2:3],
                           a[..., 0:1] * b[..., 1:2] - a[..., 1:2] * b[..., 0:1]], axis=-1)


def q_mul_vec(q, x):                  # :128-130
    t = 2.0 * cross(q[..., 1:], x)
    return x + q[..., 0][..., np.newaxis] * t + cross(q[..., 1:], t)


def to_xform_xy(x):                   # :42-55
    qw, qx, qy, qz = (x[..., i:i + 1] for i in range(4))
    x2, y2, z2 = qx + qx, qy + qy, qz + qz
    xx, yy, wx = qx * x2, qy * y2, qw * x2
    xy, yz, wy = qx * y2, qy * z2, qw * y2
    xz, zz, wz = qx * z2, qz * z2, qw * z2
    return np.concatenate([np.concatenate([1.0 - (yy + zz), xy - wz], axis=-1)[..., np.newaxis, :],
                           np.concatenate([xy + wz, 1.0 - (xx + zz)], axis=-1)[..., np.newaxis, :],
                           np.concatenate([xz - wy, yz + wx], axis=-1)[..., np.newaxis, :]], axis=-2)


def fk_vel(lrot, lpos, lvel, lang, parents):     # :189-204
    gp, gr, gv, ga = [lpos[..., :1, :]], [lrot[..., :1, :]], [lvel[..., :1, :]], [lang[..., :1, :]]
    for i in range(1, len(parents)):
        p = parents[i]
        gp.append(q_mul_vec(gr[p], lpos[..., i:i + 1, :]) + gp[p])
        gr.append(q_mul(gr[p], lrot[..., i:i + 1, :]))
        gv.append(q_mul_vec(gr[p], lvel[..., i:i + 1, :]) + cross(ga[p], q_mul_vec(gr[p], lpos[..., i:i + 1, :])) + gv[p])
        ga.append(q_mul_vec(gr[p], lang[..., i:i + 1, :]) + ga[p])
    return (np.concatenate(gr, axis=-2), np.concatenate(gp, axis=-2), np.concatenate(gv, axis=-2), np.concatenate(ga, axis=-2))


def full_parents(parents_no_root):
    """test_fullframework.py:101-102: parents = [-1] + (cfg parents + 1)."""
    return np.concatenate([[-1], np.asarray(parents_no_root) + 1])


def featurize(Yrot, Ypos, Yvel, Yang, parents):
    """Windows of local bone features (B, T, J, 4|3|3|3) float32 -> X (B, T, J, 15), un-normalised, root bone
    included (test_fullframework.py:145-185)."""
    window = Yrot.shape[1]
    Grot, Gpos, Gvel, Gang = fk_vel(Yrot, Ypos, Yvel, Yang, parents)
    Gpos[:, :, 0:1] = np.repeat(Gpos[:, -1:, 0:1], window, axis=1)
    Grot[:, :, 0:1] = np.repeat(Grot[:, -1:, 0:1], window, axis=1)
    Gvel[:, :, 0:1] = np.repeat(Gvel[:, -1:, 0:1], window, axis=1)
    Gang[:, :, 0:1] = np.repeat(Gang[:, -1:, 0:1], window, axis=1)
    inv_root = q_inv(Grot[:, :, 0:1])
    Xpos = q_mul_vec(inv_root, Gpos - Gpos[:, :, 0:1])
    Xrot = q_mul(inv_root, Grot)
    Xtxy = to_xform_xy(Xrot).astype(np.float32)
    Xvel = q_mul_vec(inv_root, Gvel)
    Xang = q_mul_vec(inv_root, Gang)
    b, ns, nj = Xtxy.shape[:3]
    return np.concatenate([Xpos, Xtxy.reshape(b, ns, nj, -1), Xvel, Xang], axis=-1).astype(np.float32)
