// Post-processing of decoded windows (test_fullframework.py:303-308, 338-437, 457-632, 665-697).
//
//  mocha_pose_heads    one 64-lane workgroup per window: last-frame pose (positions, 6-D rotation -> quaternion as
//                      motion/quat.py:96-107 / :69-94, velocities, angular velocities) and the mean hip speed over
//                      the window; float32 like the reference's NumPy on the float32 network output.
//  mocha_post_clip     the sequential frame loop of one clip per LANE (clips are the data-parallel axis): root
//                      integration, position blending, the foot-lock state machine (motion/Inertialization.py:300-377)
//                      and the two-bone IK (motion/quat.py:295-343), then the root merge + Euler channels the BVH
//                      writer consumes.  float64, because the reference's state arrays are NumPy float64.
//
// The previous frame's pose is read back from the output arrays, so the per-lane state is only the root transform and
// the contact records.
#include "kernels.h"

namespace mocha {

// ----------------------------------------------------------------------------------------------- heads
__device__ inline void cross3f(const float* a, const float* b, float* o) {
    o[0] = a[1] * b[2] - a[2] * b[1];
    o[1] = a[2] * b[0] - a[0] * b[2];
    o[2] = a[0] * b[1] - a[1] * b[0];
}

__global__ __launch_bounds__(64) void mocha_pose_heads(const float* __restrict__ Y, float* __restrict__ heads,
                                                       float* __restrict__ speed, int T, int V) {
    const int w = blockIdx.x, lane = threadIdx.x;
    const float* Yw = Y + (size_t)w * T * V * 15;
    float s = 0.f;
    for (int t = lane; t < T; t += 64) {
        const float* v = Yw + (size_t)t * V * 15 + 9;
        s += sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) speed[w] = s / (float)T;
    for (int j = lane; j < V; j += 64) {
        const float* f = Yw + ((size_t)(T - 1) * V + j) * 15;
        float* h = heads + ((size_t)w * V + j) * 13;
        // x (3,2) row-major: column 0 = f[3], f[5], f[7]; column 1 = f[4], f[6], f[8]
        const float c0[3] = {f[3], f[5], f[7]}, x1[3] = {f[4], f[6], f[8]};
        float c2[3], c1[3];
        cross3f(c0, x1, c2);
        float n = sqrtf(c2[0] * c2[0] + c2[1] * c2[1] + c2[2] * c2[2]);
        c2[0] /= n; c2[1] /= n; c2[2] /= n;
        cross3f(c2, c0, c1);
        n = sqrtf(c1[0] * c1[0] + c1[1] * c1[1] + c1[2] * c1[2]);
        c1[0] /= n; c1[1] /= n; c1[2] /= n;
        // ts[r][c]: columns c0, c1, c2
        const float t00 = c0[0], t10 = c0[1], t20 = c0[2], t01 = c1[0], t11 = c1[1], t21 = c1[2], t02 = c2[0], t12 = c2[1], t22 = c2[2];
        float q[4];
        if (t22 < 0.f) {
            if (t00 > t11) { q[0] = t21 - t12; q[1] = 1.f + t00 - t11 - t22; q[2] = t10 + t01; q[3] = t02 + t20; }
            else           { q[0] = t02 - t20; q[1] = t10 + t01; q[2] = 1.f - t00 + t11 - t22; q[3] = t21 + t12; }
        } else {
            if (t00 < -t11) { q[0] = t10 - t01; q[1] = t02 + t20; q[2] = t21 + t12; q[3] = 1.f - t00 - t11 + t22; }
            else            { q[0] = 1.f + t00 + t11 + t22; q[1] = t21 - t12; q[2] = t02 - t20; q[3] = t10 - t01; }
        }
        const float ql = sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]) + 1e-8f;
        h[0] = f[0]; h[1] = f[1]; h[2] = f[2];
        h[3] = q[0] / ql; h[4] = q[1] / ql; h[5] = q[2] / ql; h[6] = q[3] / ql;
        h[7] = f[9]; h[8] = f[10]; h[9] = f[11];
        h[10] = f[12]; h[11] = f[13]; h[12] = f[14];
    }
}

hipError_t launch_pose_heads(const float* Y, float* heads, float* speed, int B, int T, int V, hipStream_t s) {
    if (B <= 0) return hipSuccess;
    hipLaunchKernelGGL(mocha_pose_heads, dim3(B), dim3(64), 0, s, Y, heads, speed, T, V);
    return hipGetLastError();
}

// ----------------------------------------------------------------------------------------------- clip loop (float64)
struct d3 { double x, y, z; };
struct dq { double w, x, y, z; };

__device__ inline d3 operator+(d3 a, d3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ inline d3 operator-(d3 a, d3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ inline d3 operator*(d3 a, double s) { return {a.x * s, a.y * s, a.z * s}; }
__device__ inline d3 operator*(double s, d3 a) { return {a.x * s, a.y * s, a.z * s}; }
__device__ inline double dot(d3 a, d3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ inline d3 cross(d3 a, d3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
__device__ inline double len(d3 a) { return sqrt(dot(a, a)); }
__device__ inline d3 normalize(d3 a) { return a * (1.0 / (len(a) + 1e-8)); }                  // quat.py:15-16 (x / (|x| + eps))
__device__ inline dq qmul(dq x, dq y) {                                                        // quat.py:112-120
    return {y.w * x.w - y.x * x.x - y.y * x.y - y.z * x.z, y.w * x.x + y.x * x.w - y.y * x.z + y.z * x.y,
            y.w * x.y + y.x * x.z + y.y * x.w - y.z * x.x, y.w * x.z - y.x * x.y + y.y * x.x + y.z * x.w};
}
__device__ inline dq qinv(dq q) { return {q.w, -q.x, -q.y, -q.z}; }
__device__ inline d3 qrot(dq q, d3 v) {                                                        // quat.py:128-130
    const d3 u = {q.x, q.y, q.z};
    const d3 t = 2.0 * cross(u, v);
    return v + q.w * t + cross(u, t);
}
__device__ inline dq from_angle_axis(double angle, d3 axis) {                                  // quat.py:21-25
    const double c = cos(angle / 2.0), s = sin(angle / 2.0);
    return {c, s * axis.x, s * axis.y, s * axis.z};
}
__device__ inline dq from_scaled_angle_axis(d3 v) {                                            // quat.py:154-164
    const d3 x = v * 0.5;
    const double h = len(x);
    if (h < 1e-5) return {1.0, x.x, x.y, x.z};
    const double s = sin(h) / h;
    return {cos(h), s * x.x, s * x.y, s * x.z};
}
__device__ inline double clamp1(double x) { return x < -1.0 ? -1.0 : (x > 1.0 ? 1.0 : x); }

__device__ inline d3 ld3(const double* p) { return {p[0], p[1], p[2]}; }
__device__ inline dq ldq(const double* p) { return {p[0], p[1], p[2], p[3]}; }
__device__ inline void st3(double* p, d3 v) { p[0] = v.x; p[1] = v.y; p[2] = v.z; }
__device__ inline void stq(double* p, dq q) { p[0] = q.w; p[1] = q.x; p[2] = q.y; p[3] = q.z; }

struct ContactRec {
    bool state, lock;
    d3 position, velocity, point, target, off_x, off_v;
};

// One WAVE per clip (round 3; the first version ran a clip in one lane, 64 clips per wave): the frame loop is sequential, but inside a
// frame the work has width - lane j owns bone j (its blended position is a per-bone recurrence over frames, kept in a register), lane 0
// also integrates the root, lane 32 + c runs contact c's state machine and two-bone IK - and everything a frame needs from memory (the
// bone's 13 head values, the root's source velocities, the contact flags) is fetched one frame AHEAD, so the per-frame critical path is
// the float64 arithmetic itself: root integration, then the chain and the IK of a contact, with one exchange through LDS between them.
// Every value is computed by the same expressions in the same order as before (and as the oracle): results are unchanged.
__global__ __launch_bounds__(64) void mocha_post_clip(PostParams p) {
    __shared__ double sh_pos[MOCHA_MAX_BONES * 3], sh_rot[MOCHA_MAX_BONES * 4];
    __shared__ double sh_w[6];                                   // wv, wa of the frame (the contact reset of frame 0 needs them)
    __shared__ double sh_ikq[MOCHA_MAX_CONTACT][2][4];
    __shared__ int sh_ikv[MOCHA_MAX_CONTACT];
    const int clip = blockIdx.x, lane = threadIdx.x;
    const int V = p.V, J = V + 1, N = p.n_frames;
    const float* heads = p.heads + (size_t)clip * N * V * 13;
    const float* speed = p.speed + (size_t)clip * N;
    const float* rvel_s = p.src_rvel + (size_t)clip * N * 3;
    const float* rang_s = p.src_rang + (size_t)clip * N * 3;
    const float* sspeed = p.src_speed + (size_t)clip * N;
    const unsigned char* contact = p.contact + (size_t)clip * N * p.n_contact;
    double* POS = p.pos + (size_t)clip * N * J * 3;
    double* ROT = p.rot + (size_t)clip * N * J * 4;
    double* IKR = p.ik_rot + (size_t)clip * N * J * 4;
    double* BP = (p.bvh_pos && p.bvh_euler) ? p.bvh_pos + (size_t)clip * N * V * 3 : nullptr;
    const double dt = p.dt;
    const double ydamp = (4.0 * 0.6931471805599453) / (p.halflife + 1e-5) / 2.0;             // Inertialization.py:13-14, 40
    const double eydt = 1.0 / (1.0 + ydamp * dt + 0.48 * (ydamp * dt) * (ydamp * dt) + 0.235 * (ydamp * dt) * (ydamp * dt) * (ydamp * dt));

    const bool is_joint = lane < J, is_bone = lane >= 1 && lane < J;
    const int ci = lane - 32;
    const bool is_contact = ci >= 0 && ci < p.n_contact;
    // a contact lane's chain toe -> ... -> root (quat.py:241-273 recursion order, evaluated top-down), fixed for the clip
    int chain[MOCHA_MAX_CHAIN], depth = 0;
    if (is_contact)
        for (int b = p.contact_bones[ci]; b != -1 && depth < MOCHA_MAX_CHAIN; b = p.parents[b]) chain[depth++] = b;
    // which contacts write this bone's IK rotation (chain[3] the hip, chain[2] the knee), in contact order
    int ik_slot[MOCHA_MAX_CONTACT];
#pragma unroll
    for (int c = 0; c < MOCHA_MAX_CONTACT; ++c) {
        ik_slot[c] = -1;
        if (c < p.n_contact && is_joint) {
            int ch[MOCHA_MAX_CHAIN], d = 0;
            for (int b = p.contact_bones[c]; b != -1 && d < MOCHA_MAX_CHAIN; b = p.parents[b]) ch[d++] = b;
            if (d >= 5) { if (ch[3] == lane) ik_slot[c] = 0; else if (ch[2] == lane) ik_slot[c] = 1; }
        }
    }

    // per-lane state
    ContactRec cr;                                              // contact lanes
    d3 root_pos = {0, 0, 0};                                    // lane 0
    dq root_rot = {1, 0, 0, 0};
    d3 prev_p = {0, 0, 0};                                      // joint lanes: this bone's position in the previous frame
    // inputs of the frame about to be processed, fetched one frame ahead
    float hn[13];                                               // bone lanes
    float r_speed = 0.f, r_sspeed = 1.f, r_rv[3] = {0, 0, 0}, r_ra[3] = {0, 0, 0};            // lane 0
    unsigned char c_flag = 0;                                   // contact lanes
    auto fetch = [&](int i) __attribute__((always_inline)) {
        if (i >= N) return;
        if (is_bone) {
            const float* hj = heads + ((size_t)i * V + (lane - 1)) * 13;
#pragma unroll
            for (int k = 0; k < 13; ++k) hn[k] = hj[k];
        }
        if (lane == 0) {
            r_speed = speed[i]; r_sspeed = sspeed[i];
            r_rv[0] = rvel_s[i * 3]; r_rv[1] = rvel_s[i * 3 + 1]; r_rv[2] = rvel_s[i * 3 + 2];
            r_ra[0] = rang_s[i * 3]; r_ra[1] = rang_s[i * 3 + 1]; r_ra[2] = rang_s[i * 3 + 2];
        }
        if (is_contact) c_flag = contact[(size_t)i * p.n_contact + ci];
    };
    fetch(0);

    for (int i = 0; i < N; ++i) {
        // this frame's inputs into locals, the next frame's on their way
        float h[13];
#pragma unroll
        for (int k = 0; k < 13; ++k) h[k] = hn[k];
        const float f_speed = r_speed, f_sspeed = r_sspeed;
        const float f_rv[3] = {r_rv[0], r_rv[1], r_rv[2]}, f_ra[3] = {r_ra[0], r_ra[1], r_ra[2]};
        const bool in_state = c_flag != 0;
        fetch(i + 1);

        double* pos = POS + (size_t)i * J * 3;
        double* rot = ROT + (size_t)i * J * 4;
        double* ikr = IKR + (size_t)i * J * 4;
        d3 pj = {0, 0, 0}, vj = {0, 0, 0};
        dq rj = {1, 0, 0, 0};
        if (lane == 0) {
            // root-velocity ratio in float32, as NumPy computes it on float32 arrays (test_fullframework.py:492-496)
            float ratio = f_speed / f_sspeed;
            if (ratio > 3.0f || ratio < 0.33f) ratio = 1.0f;
            const d3 rv = {(double)(f_rv[0] * ratio), (double)(f_rv[1] * ratio), (double)(f_rv[2] * ratio)};
            const d3 ra = {(double)f_ra[0], (double)f_ra[1], (double)f_ra[2]};
            const d3 wv = qrot(root_rot, rv), wa = qrot(root_rot, ra);                         // :499-502
            root_pos = root_pos + wv * dt;
            root_rot = qmul(root_rot, from_scaled_angle_axis(wa * dt));
            pj = root_pos; vj = wv; rj = root_rot;
            st3(sh_w, wv); st3(sh_w + 3, wa);
            if (BP) st3(BP + (size_t)i * V * 3, root_pos);          // for mocha_post_bvh: the running (unblended) root of this frame
        } else if (is_bone) {
            pj = {(double)h[0], (double)h[1], (double)h[2]};
            rj = {(double)h[3], (double)h[4], (double)h[5], (double)h[6]};
            vj = {(double)h[7], (double)h[8], (double)h[9]};
        }
        if (is_joint) {
            // positions (blended with the previous frame's after the first one, :537/:627), rotations
            if (i && p.blend_enabled) pj = (prev_p + vj * dt) * 0.5 + pj * 0.5;
            prev_p = pj;
            st3(pos + lane * 3, pj);
            stq(rot + lane * 4, rj);
            st3(sh_pos + lane * 3, pj);
            stq(sh_rot + lane * 4, rj);
        }
        if (lane < MOCHA_MAX_CONTACT) sh_ikv[lane] = 0;
        __syncthreads();

        if (is_contact) {
            if (i == 0) {
                // contact reset with the global position / velocity of the toe (fk_vel_bone, quat.py:207-238)
                const d3 wv = ld3(sh_w), wa = ld3(sh_w + 3);
                d3 gv = {0, 0, 0}, ga = {0, 0, 0}, gpp = {0, 0, 0};
                dq grr = {1, 0, 0, 0};
                for (int k = depth - 1; k >= 0; --k) {
                    const int b = chain[k];
                    d3 lv, la;
                    if (b == 0) { lv = wv; la = wa; }
                    else {
                        const float* hb = heads + (size_t)(b - 1) * 13;                        // frame 0
                        lv = {(double)hb[7], (double)hb[8], (double)hb[9]};
                        la = {(double)hb[10], (double)hb[11], (double)hb[12]};
                    }
                    const d3 lp = ld3(sh_pos + b * 3);
                    const dq lr = ldq(sh_rot + b * 4);
                    if (k == depth - 1) { gpp = lp; gv = lv; grr = lr; ga = la; }
                    else {
                        const d3 rp = qrot(grr, lp);
                        gv = gv + qrot(grr, lv) + cross(ga, rp);
                        ga = qrot(grr, la) + ga;
                        gpp = rp + gpp;
                        grr = qmul(grr, lr);
                    }
                }
                cr.state = false; cr.lock = false;
                cr.position = gpp; cr.velocity = gv; cr.point = gpp; cr.target = gpp;
                cr.off_x = {0, 0, 0}; cr.off_v = {0, 0, 0};
            } else if (p.ik_enabled && depth >= 5) {
                d3 gp[MOCHA_MAX_CHAIN];
                dq gr[MOCHA_MAX_CHAIN];
                for (int k = depth - 1; k >= 0; --k) {
                    const int b = chain[k];
                    const d3 lp = ld3(sh_pos + b * 3);
                    const dq lr = ldq(sh_rot + b * 4);
                    if (k == depth - 1) { gp[k] = lp; gr[k] = lr; }
                    else { gp[k] = qrot(gr[k + 1], lp) + gp[k + 1]; gr[k] = qmul(gr[k + 1], lr); }
                }
                // chain[0] toe, [1] heel, [2] knee, [3] hip, [4] the hip's parent
                ContactRec& c = cr;
                const d3 in_pos = gp[0];
                // Inertialization.py:300-377
                const d3 in_vel = (in_pos - c.target) * (1.0 / (dt + 1e-8));
                c.target = in_pos;
                {
                    const d3 j1 = c.off_v + c.off_x * ydamp;                                   // :39-54
                    c.off_x = eydt * (c.off_x + j1 * dt);
                    c.off_v = eydt * (c.off_v - j1 * (ydamp * dt));
                }
                if (c.lock) { c.position = c.point + c.off_x; c.velocity = c.off_v; }
                else { c.position = in_pos + c.off_x; c.velocity = in_vel + c.off_v; }
                const bool unlock = c.lock && len(c.point - in_pos) > p.unlock_radius;
                if (!c.state && in_state) {
                    c.lock = true;
                    c.point = c.position;
                    c.point.y = p.foot_height;
                    c.off_x = (in_pos + c.off_x) - c.point;
                    c.off_v = in_vel + c.off_v;
                } else if ((c.lock && c.state && !in_state) || unlock) {
                    c.lock = false;
                    c.off_x = (c.point + c.off_x) - in_pos;
                    c.off_v = c.off_v - in_vel;
                }
                c.state = in_state;
                if (c.position.y < p.foot_height) c.position.y = p.foot_height;                // test_fullframework.py:581-582
                // two-bone IK (quat.py:295-343): a hip, b knee, c heel
                const d3 a = gp[3], b = gp[2], e = gp[1];
                const d3 target = c.position + (gp[1] - gp[0]);
                const d3 fwd = qrot(gr[2], d3{0.0, 1.0, 0.0});
                const double max_ext = len(a - b) + len(b - e) - p.max_length_buffer;
                d3 t = target;
                if (len(target - a) > max_ext) t = a + max_ext * normalize(target - a);
                const d3 axis_rot = normalize(cross(normalize(e - a), fwd));
                const double lab = len(b - a), lcb = len(b - e), lat = len(t - a);
                const double ac_ab_0 = acos(clamp1(dot(normalize(e - a), normalize(b - a))));
                const double ba_bc_0 = acos(clamp1(dot(normalize(a - b), normalize(e - b))));
                const double ac_ab_1 = acos(clamp1((lab * lab + lat * lat - lcb * lcb) / (2.0 * lab * lat)));
                const double ba_bc_1 = acos(clamp1((lab * lab + lcb * lcb - lat * lat) / (2.0 * lab * lcb)));
                const dq r0 = from_angle_axis(ac_ab_1 - ac_ab_0, axis_rot);
                const dq r1 = from_angle_axis(ba_bc_1 - ba_bc_0, axis_rot);
                const d3 c_a = normalize(e - a), t_a = normalize(t - a);
                const dq r2 = from_angle_axis(acos(clamp1(dot(c_a, t_a))), normalize(cross(c_a, t_a)));
                stq(&sh_ikq[ci][0][0], qmul(qinv(gr[4]), qmul(r2, qmul(r0, gr[3]))));          // the hip's (chain[3])
                stq(&sh_ikq[ci][1][0], qmul(qinv(gr[3]), qmul(r1, gr[2])));                    // the knee's (chain[2])
                sh_ikv[ci] = 1;
            }
        }
        __syncthreads();
        if (is_joint) {
            // the IK rotations: the bone's own rotation unless a contact replaced it (later contacts win, as in the sequential order)
            dq q = rj;
#pragma unroll
            for (int c = 0; c < MOCHA_MAX_CONTACT; ++c)
                if (ik_slot[c] >= 0 && sh_ikv[c]) q = ldq(&sh_ikq[c][ik_slot[c]][0]);
            stq(ikr + lane * 4, q);
        }
        __syncthreads();                                        // the LDS exchange is rewritten by the next frame
    }
}

// BVH channels (test_fullframework.py:677-681; quat.py:346-355): fold the synthetic root into bone 1, Euler angles in degrees.  No
// dependence between frames or bones - 3 float64 inverse trigonometric functions per bone, which the frame loop above used to
// evaluate one after the other in its single lane (72 per frame: most of its 46 us per frame for one clip).
__global__ __launch_bounds__(256) void mocha_post_bvh(PostParams p) {
    const int V = p.V, J = V + 1, N = p.n_frames;
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= (long long)p.n_clips * N * V) return;
    const int j = (int)(t % V) + 1;
    const long long ci = t / V;                     // clip * N + frame
    const double* pos = p.pos + (size_t)ci * J * 3;
    const double* ikr = p.ik_rot + (size_t)ci * J * 4;
    double* bp = p.bvh_pos + ((size_t)ci * V + (j - 1)) * 3;
    d3 pj = ld3(pos + j * 3);
    dq q = ldq(ikr + j * 4);
    if (j == 1) {
        const d3 root_pos = ld3(bp);                // parked by the frame loop
        const dq root_rot = ldq(p.rot + (size_t)ci * J * 4);       // ROT[frame][0]: the running root rotation (rotations are not blended)
        pj = qrot(root_rot, pj) + root_pos; q = qmul(root_rot, q);
    }
    st3(bp, pj);
    const double r2d = 57.29577951308232;
    d3 eu = {atan2(2 * (q.w * q.x + q.y * q.z), 1 - 2 * (q.x * q.x + q.y * q.y)), asin(clamp1(2 * (q.w * q.y - q.z * q.x))),
             atan2(2 * (q.w * q.z + q.x * q.y), 1 - 2 * (q.y * q.y + q.z * q.z))};
    st3(p.bvh_euler + ((size_t)ci * V + (j - 1)) * 3, eu * r2d);
}

hipError_t launch_post_clip(const PostParams& p, hipStream_t s) {
    if (p.n_clips <= 0 || p.n_frames <= 0) return hipSuccess;
    if (p.V + 1 > MOCHA_MAX_BONES || p.n_contact > MOCHA_MAX_CONTACT) return hipErrorInvalidValue;
    hipLaunchKernelGGL(mocha_post_clip, dim3(p.n_clips), dim3(64), 0, s, p);
    if (p.bvh_pos && p.bvh_euler) {
        const long long n = (long long)p.n_clips * p.n_frames * p.V;
        hipLaunchKernelGGL(mocha_post_bvh, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, p);
    }
    return hipGetLastError();
}

}  // namespace mocha
