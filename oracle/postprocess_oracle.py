"""CPU ORACLE for the demo's per-frame post-processing — TEST INFRASTRUCTURE.

NumPy restatement of what test_fullframework.py does with every decoded window after ``to_mot``:

* ``pose_heads``      — last-frame pose of a de-normalised window (:303-308, 457-462): positions, 6-D rotation ->
  quaternion (motion/quat.py:96-107 ``from_xform_xy`` / :69-94 ``from_xform``), velocities, angular velocities, and
  the mean hip speed over the window used by the root-velocity ratio (:338-339, 492-493).
* ``PostProcess``     — the sequential state of the frame loop: root integration (:345-352, 499-508), position
  blending (:537, 627), the foot-lock state machine (motion/Inertialization.py:300-377 with its spring helpers
  :10-16, 39-68, 93-133) and the two-bone IK (motion/quat.py:241-273 ``fk_partial``, :295-343 ``ik_two_bone``).
* ``bvh_channels`` / ``write_bvh`` — root merge + Euler conversion (:665-687, motion/quat.py:346-358) and the BVH
  text layout (motion/bvh.py:145-224).

The reference mixes float32 network outputs with float64 NumPy state (its zeros()/array([1,0,0,0]) are float64), so
everything after ``pose_heads`` is float64 here too.  Pinned against the reference's own quat / Inertialization
modules driven in the demo's order on synthetic inputs (tests/golden/postprocess.npz)."""
from __future__ import annotations

import numpy as np

DT = 1.0 / 60.0                     # test_fullframework.py:105
IK = dict(max_length_buffer=0.015, foot_height=0.02, unlock_radius=0.2, blending_halflife=0.1)   # :109-114
CONTACT_BONES = (5, 24)             # :104


# ----------------------------------------------------------------------------- quaternion helpers (motion/quat.py)
def cross(a, b):
    return np.array([a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]])


def length(x):
    return np.sqrt(np.sum(x * x))


def normalize(x, eps=1e-8):         # :15-16
    return x / (length(x) + eps)


def q_mul(x, y):                    # :112-120
    return np.array([y[0] * x[0] - y[1] * x[1] - y[2] * x[2] - y[3] * x[3],
                     y[0] * x[1] + y[1] * x[0] - y[2] * x[3] + y[3] * x[2],
                     y[0] * x[2] + y[1] * x[3] + y[2] * x[0] - y[3] * x[1],
                     y[0] * x[3] - y[1] * x[2] + y[2] * x[1] + y[3] * x[0]])


def q_inv(q):
    return np.array([q[0], -q[1], -q[2], -q[3]])


def q_mul_vec(q, x):                # :128-130
    t = 2.0 * cross(q[1:], x)
    return x + q[0] * t + cross(q[1:], t)


def q_exp(x, eps=1e-5):             # :154-158
    h = np.sqrt(np.sum(x * x))
    if h < eps:
        return np.concatenate([[1.0], x])
    return np.concatenate([[np.cos(h)], np.sin(h) / h * x])


def from_scaled_angle_axis(x):      # :163-164
    return q_exp(x / 2.0)


def from_angle_axis(angle, axis):   # :21-25
    return np.concatenate([[np.cos(angle / 2.0)], np.sin(angle / 2.0) * axis])


def between(x, y):                  # :143-147
    return np.concatenate([[np.sqrt(np.sum(x * x) * np.sum(y * y)) + np.sum(x * y)], cross(x, y)])


def from_xform(ts):                 # :69-94, vectorised over leading axes, keeps the input dtype
    one = ts.dtype.type(1.0)
    t00, t11, t22 = ts[..., 0, 0], ts[..., 1, 1], ts[..., 2, 2]
    a = np.stack([ts[..., 2, 1] - ts[..., 1, 2], one + t00 - t11 - t22, ts[..., 1, 0] + ts[..., 0, 1], ts[..., 0, 2] + ts[..., 2, 0]], -1)
    b = np.stack([ts[..., 0, 2] - ts[..., 2, 0], ts[..., 1, 0] + ts[..., 0, 1], one - t00 + t11 - t22, ts[..., 2, 1] + ts[..., 1, 2]], -1)
    c = np.stack([ts[..., 1, 0] - ts[..., 0, 1], ts[..., 0, 2] + ts[..., 2, 0], ts[..., 2, 1] + ts[..., 1, 2], one - t00 - t11 + t22], -1)
    d = np.stack([one + t00 + t11 + t22, ts[..., 2, 1] - ts[..., 1, 2], ts[..., 0, 2] - ts[..., 2, 0], ts[..., 1, 0] - ts[..., 0, 1]], -1)
    q = np.where((t22 < 0)[..., None], np.where((t00 > t11)[..., None], a, b), np.where((t00 < -t11)[..., None], c, d))
    return q / (np.sqrt(np.sum(q * q, -1))[..., None] + ts.dtype.type(1e-8))


def from_xform_xy(x):               # :96-107; x (..., 3, 2)
    def vcross(a, b):
        return np.stack([a[..., 1] * b[..., 2] - a[..., 2] * b[..., 1], a[..., 2] * b[..., 0] - a[..., 0] * b[..., 2],
                         a[..., 0] * b[..., 1] - a[..., 1] * b[..., 0]], -1)
    c2 = vcross(x[..., 0], x[..., 1])
    c2 = c2 / np.sqrt(np.sum(np.square(c2), -1))[..., None]
    c1 = vcross(c2, x[..., 0])
    c1 = c1 / np.sqrt(np.sum(np.square(c1), -1))[..., None]
    return from_xform(np.stack([x[..., 0], c1, c2], -1))


def to_euler_xyz(x):                # :346-355 (the demo calls to_euler with its default order)
    q0, q1, q2, q3 = (x[..., i] for i in range(4))
    return np.stack([np.arctan2(2 * (q0 * q1 + q2 * q3), 1 - 2 * (q1 * q1 + q2 * q2)),
                     np.arcsin((2 * (q0 * q2 - q3 * q1)).clip(-1, 1)),
                     np.arctan2(2 * (q0 * q3 + q1 * q2), 1 - 2 * (q2 * q2 + q3 * q3))], -1)


# ----------------------------------------------------------------------------- heads of a decoded window
def pose_heads(Y):
    """Y (B,60,V,15) de-normalised float32 -> heads (B,V,13) = [pos 3 | quat 4 | vel 3 | ang 3] of the LAST frame and
    speed (B,) = mean_t |Y[t, 0, 9:12]|   (test_fullframework.py:304-308, 338)."""
    Y = np.asarray(Y, np.float32)
    last = Y[:, -1]
    rot = from_xform_xy(last[..., 3:9].reshape(last.shape[0], last.shape[1], 3, 2))
    heads = np.concatenate([last[..., :3], rot, last[..., 9:12], last[..., 12:15]], -1).astype(np.float32)
    speed = np.linalg.norm(Y[:, :, 0, 9:12], axis=-1).mean(-1)
    return heads, speed.astype(np.float32)


# ----------------------------------------------------------------------------- foot lock (motion/Inertialization.py)
def _fast_negexp(x):                # :10-11
    return 1.0 / (1.0 + x + 0.48 * x * x + 0.235 * x * x * x)


def _decay(x, v, halflife, dt):     # :39-54 (vector branch), :13-14
    y = (4.0 * np.log(2.0)) / (halflife + 1e-5) / 2.0
    e = _fast_negexp(y * dt)
    j1 = v + x * y
    return e * (x + j1 * dt), e * (v - j1 * y * dt)


class Contact:
    """State of one contact bone (test_fullframework.py:402-430)."""

    def __init__(self, pos, vel):
        self.state = False; self.lock = False
        self.position = pos.copy(); self.velocity = vel.copy()
        self.point = pos.copy(); self.target = pos.copy()
        self.off_x = np.zeros(3); self.off_v = np.zeros(3)

    def update(self, in_pos, in_state, unlock_radius, foot_height, halflife, dt, eps=1e-8):   # Inertialization.py:300-377
        in_vel = (in_pos - self.target) / (dt + eps)
        self.target = in_pos.copy()
        self.off_x, self.off_v = _decay(self.off_x, self.off_v, halflife, dt)                     # :110-116
        feed_x, feed_v = (self.point, np.zeros(3)) if self.lock else (in_pos, in_vel)
        self.position, self.velocity = feed_x + self.off_x, feed_v + self.off_v
        unlock = self.lock and length(self.point - in_pos) > unlock_radius
        if (not self.state) and in_state:
            self.lock = True
            self.point = self.position.copy(); self.point[1] = foot_height
            self.off_x, self.off_v = (in_pos + self.off_x) - self.point, (in_vel + self.off_v) - 0.0   # :93-98
        elif (self.lock and self.state and not in_state) or unlock:
            self.lock = False
            self.off_x, self.off_v = (self.point + self.off_x) - in_pos, (0.0 + self.off_v) - in_vel
        self.state = bool(in_state)


def ik_two_bone(root_lr, mid_lr, a, b, c, target, fwd, root_gr, mid_gr, par_gr, max_length_buffer):   # quat.py:295-343
    max_ext = length(a - b) + length(b - c) - max_length_buffer
    t = target
    if length(target - a) > max_ext:
        t = a + max_ext * normalize(target - a)
    axis_rot = normalize(np.cross(normalize(c - a), fwd))
    lab, lcb, lat = length(b - a), length(b - c), length(t - a)
    ac_ab_0 = np.arccos(np.clip(np.dot(normalize(c - a), normalize(b - a)), -1.0, 1.0))
    ba_bc_0 = np.arccos(np.clip(np.dot(normalize(a - b), normalize(c - b)), -1.0, 1.0))
    ac_ab_1 = np.arccos(np.clip((lab * lab + lat * lat - lcb * lcb) / (2.0 * lab * lat), -1.0, 1.0))
    ba_bc_1 = np.arccos(np.clip((lab * lab + lcb * lcb - lat * lat) / (2.0 * lab * lcb), -1.0, 1.0))
    r0 = from_angle_axis(ac_ab_1 - ac_ab_0, axis_rot)
    r1 = from_angle_axis(ba_bc_1 - ba_bc_0, axis_rot)
    c_a, t_a = normalize(c - a), normalize(t - a)
    r2 = from_angle_axis(np.arccos(np.clip(np.dot(c_a, t_a), -1.0, 1.0)), normalize(np.cross(c_a, t_a)))
    return q_mul(q_inv(par_gr), q_mul(r2, q_mul(r0, root_gr))), q_mul(q_inv(root_gr), q_mul(r1, mid_gr))


def _fk_chain(pos, rot, parents, bone):
    """Global position / rotation of ``bone`` and of all its ancestors (quat.py:241-273 without the memo flags)."""
    chain = []
    b = bone
    while b != -1:
        chain.append(b); b = parents[b]
    gp, gr = {}, {}
    for b in reversed(chain):
        p = parents[b]
        if p == -1:
            gp[b], gr[b] = pos[b].astype(np.float64), rot[b].astype(np.float64)
        else:
            gp[b], gr[b] = q_mul_vec(gr[p], pos[b]) + gp[p], q_mul(gr[p], rot[b])
    return gp, gr


def _fk_vel_bone(pos, vel, rot, ang, parents, bone):     # quat.py:207-238
    if parents[bone] == -1:
        return pos[bone], vel[bone], rot[bone], ang[bone]
    pp, pv, pr, pa = _fk_vel_bone(pos, vel, rot, ang, parents, parents[bone])
    return (q_mul_vec(pr, pos[bone]) + pp, pv + q_mul_vec(pr, vel[bone]) + cross(pa, q_mul_vec(pr, pos[bone])),
            q_mul(pr, rot[bone]), q_mul_vec(pr, ang[bone]) + pa)


class PostProcess:
    """Sequential per-frame state of one clip.  ``step`` consumes the heads of one decoded window plus the source's
    root-local velocities / hip speed / contact labels of the same frame and returns the frame's pose:
    (pos (V+1,3), rot (V+1,4), ik_rot (V+1,4)), all float64; pos is shared by the plain and the IK stream."""

    def __init__(self, parents, contact_bones=CONTACT_BONES, dt=DT, ik=IK, ik_enabled=True, blend=True):
        self.parents = [int(p) for p in parents]
        self.contact_bones = tuple(int(b) for b in contact_bones)
        self.dt, self.ik, self.ik_enabled = dt, dict(ik), ik_enabled
        self.blend = blend            # False + ik_enabled=False: the demo's context-matching "cm_" stream (:512-527, 637-641)
        self.prev_pos = None

    def _root(self, heads, speed, src_rvel, src_rang, src_speed, root_pos, root_rot):
        ratio = np.float32(speed) / np.float32(src_speed)                              # :492-495, float32 like the demo
        if ratio > 3.0 or ratio < 0.33:
            ratio = np.float32(1.0)
        rvel = (np.asarray(src_rvel, np.float32) * ratio).astype(np.float64)
        rang = np.asarray(src_rang, np.float64)
        vel = q_mul_vec(root_rot, rvel); ang = q_mul_vec(root_rot, rang)               # :499-502
        pos = root_pos + vel * self.dt
        rot = q_mul(root_rot, from_scaled_angle_axis(ang * self.dt))
        h = np.asarray(heads, np.float64)
        return (np.concatenate([pos[None], h[:, 0:3]]), np.concatenate([rot[None], h[:, 3:7]]),
                np.concatenate([vel[None], h[:, 7:10]]), np.concatenate([ang[None], h[:, 10:13]]))

    def step(self, heads, speed, src_rvel, src_rang, src_speed, contact):
        dt = self.dt
        if self.prev_pos is None:                                                       # first frame, :338-437
            pos, rot, vel, ang = self._root(heads, speed, src_rvel, src_rang, src_speed, np.zeros(3), np.array([1.0, 0, 0, 0]))
            self.contacts = []
            for b in self.contact_bones:
                bp, bv, _, _ = _fk_vel_bone(pos, vel, rot, ang, self.parents, b)
                self.contacts.append(Contact(bp, bv))
            self.prev_pos, self.prev_root_rot = pos, rot[0]
            return pos.copy(), rot.copy(), rot.copy()
        pos, rot, vel, ang = self._root(heads, speed, src_rvel, src_rang, src_speed, self.prev_pos[0], self.prev_root_rot)
        blended = (self.prev_pos + vel * dt) * 0.5 + pos * 0.5 if self.blend else pos  # :537-540, 627; cm_ stream: :524, 637
        ik_rot = rot.copy()
        if self.ik_enabled:
            for ci, toe in enumerate(self.contact_bones):
                heel = self.parents[toe]; knee = self.parents[heel]; hip = self.parents[knee]; par = self.parents[hip]
                gp, gr = _fk_chain(blended, rot, self.parents, toe)                    # :551-558
                c = self.contacts[ci]
                c.update(gp[toe], bool(contact[ci]), self.ik["unlock_radius"], self.ik["foot_height"],
                         self.ik["blending_halflife"], dt)                            # :560-578
                c.position[1] = max(c.position[1], self.ik["foot_height"])            # :581-582 (in place)
                ik_rot[hip], ik_rot[knee] = ik_two_bone(                               # :596-608
                    ik_rot[hip], ik_rot[knee], gp[hip], gp[knee], gp[heel], c.position + (gp[heel] - gp[toe]),
                    q_mul_vec(gr[knee], np.array([0.0, 1.0, 0.0])), gr[hip], gr[knee], gr[par], self.ik["max_length_buffer"])
        self.prev_pos, self.prev_root_rot = blended, rot[0]
        return blended.copy(), rot.copy(), ik_rot


def run_clip(heads, speed, src_rvel, src_rang, src_speed, contact, parents, **kw):
    """All frames of one clip: heads (N,V,13) ... -> pos (N,V+1,3), rot (N,V+1,4), ik_rot (N,V+1,4)."""
    pp = PostProcess(parents, **kw)
    out = [pp.step(heads[i], speed[i], src_rvel[i], src_rang[i], src_speed[i], contact[i]) for i in range(len(heads))]
    return tuple(np.stack([o[k] for o in out]) for k in range(3))


def bvh_channels(pos, rot):
    """Fold the synthetic root into bone 1 and convert to the demo's BVH channels (test_fullframework.py:665-687):
    returns positions (N,V,3) and Euler angles in degrees (N,V,3)."""
    p, r = pos[:, 1:].copy(), rot[:, 1:].copy()
    for i in range(len(pos)):
        p[i, 0] = q_mul_vec(rot[i, 0], pos[i, 1]) + pos[i, 0]
        r[i, 0] = q_mul(rot[i, 0], rot[i, 1])
    return p, np.degrees(to_euler_xyz(r))
