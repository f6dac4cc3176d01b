// Window featurisation of the demo on the device (test_fullframework.py:141-185):
// forward kinematics with velocities over the bone tree (motion/quat.py:189-204), re-rooting of each
// window on its own last frame (:148-151), every bone expressed in that root frame (:154-158) and the
// features concatenated as [pos 3 | first two columns of the rotation matrix 6 | vel 3 | ang 3] (:180-185).
// One thread per (window, frame); the per-bone global transforms a thread needs for its children live in
// LDS ([bone][component][thread], conflict-free).  Quaternions are (w, x, y, z) as in the reference.
#include "kernels.h"
#include "device_utils.h"

namespace mocha {

struct Q { float w, x, y, z; };
struct V3 { float x, y, z; };

__device__ __forceinline__ V3 crossv(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
__device__ __forceinline__ V3 addv(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ V3 subv(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
// motion/quat.py:112-120: mul(x, y)
__device__ __forceinline__ Q qmul(Q x, Q y) {
    return {y.w * x.w - y.x * x.x - y.y * x.y - y.z * x.z,
            y.w * x.x + y.x * x.w - y.y * x.z + y.z * x.y,
            y.w * x.y + y.x * x.z + y.y * x.w - y.z * x.x,
            y.w * x.z - y.x * x.y + y.y * x.x + y.z * x.w};
}
__device__ __forceinline__ Q qinv(Q q) { return {q.w, -q.x, -q.y, -q.z}; }
// motion/quat.py:128-130: t = 2 cross(q.xyz, v); v + w t + cross(q.xyz, t)
__device__ __forceinline__ V3 qrot(Q q, V3 v) {
    const V3 u = {q.x, q.y, q.z};
    V3 t = crossv(u, v);
    t = {2.0f * t.x, 2.0f * t.y, 2.0f * t.z};
    const V3 c = crossv(u, t);
    return {v.x + q.w * t.x + c.x, v.y + q.w * t.y + c.y, v.z + q.w * t.z + c.z};
}

static constexpr int FT = 64;       // threads per workgroup
static constexpr int FC = 13;       // floats kept per bone: rot 4, pos 3, vel 3, ang 3

// no packed fp32 instructions: LDS-fed v_pk_*_f32 with op_sel is the combination that misbehaved in mocha_body_front (pointwise.hip)
__global__ __launch_bounds__(FT) MOCHA_NO_PACKED_F32 void mocha_featurize(const float* __restrict__ Yrot, const float* __restrict__ Ypos,
                                                      const float* __restrict__ Yvel, const float* __restrict__ Yang,
                                                      const int* __restrict__ parents, float* __restrict__ X, int frames /*B*T*/,
                                                      int T, int J) {
    extern __shared__ float g[];                                  // [J][FC][FT]
    const int tid = threadIdx.x;
    const int f = blockIdx.x * FT + tid;
    if (f >= frames) return;
    const int b = f / T;
    const size_t last = ((size_t)b * T + (T - 1)) * J;             // bone 0 of the window's last frame
    const Q Rr = {Yrot[last * 4], Yrot[last * 4 + 1], Yrot[last * 4 + 2], Yrot[last * 4 + 3]};
    const V3 Rp = {Ypos[last * 3], Ypos[last * 3 + 1], Ypos[last * 3 + 2]};
    const V3 Rv = {Yvel[last * 3], Yvel[last * 3 + 1], Yvel[last * 3 + 2]};
    const V3 Ra = {Yang[last * 3], Yang[last * 3 + 1], Yang[last * 3 + 2]};
    const Q Ri = qinv(Rr);
    auto G = [&](int bone, int comp) -> float& { return g[(bone * FC + comp) * FT + tid]; };

    for (int i = 0; i < J; ++i) {
        const size_t e = (size_t)f * J + i;
        const Q lr = {Yrot[e * 4], Yrot[e * 4 + 1], Yrot[e * 4 + 2], Yrot[e * 4 + 3]};
        const V3 lp = {Ypos[e * 3], Ypos[e * 3 + 1], Ypos[e * 3 + 2]};
        const V3 lv = {Yvel[e * 3], Yvel[e * 3 + 1], Yvel[e * 3 + 2]};
        const V3 la = {Yang[e * 3], Yang[e * 3 + 1], Yang[e * 3 + 2]};
        Q gr; V3 gp, gv, ga;
        if (i == 0) {
            gr = lr; gp = lp; gv = lv; ga = la;
        } else {
            const int p = parents[i];
            const Q pr = {G(p, 0), G(p, 1), G(p, 2), G(p, 3)};
            const V3 pp = {G(p, 4), G(p, 5), G(p, 6)}, pv = {G(p, 7), G(p, 8), G(p, 9)}, pa = {G(p, 10), G(p, 11), G(p, 12)};
            const V3 rp = qrot(pr, lp);
            gp = addv(rp, pp);
            gr = qmul(pr, lr);
            gv = addv(addv(qrot(pr, lv), crossv(pa, rp)), pv);
            ga = addv(qrot(pr, la), pa);
        }
        G(i, 0) = gr.w; G(i, 1) = gr.x; G(i, 2) = gr.y; G(i, 3) = gr.z;
        G(i, 4) = gp.x; G(i, 5) = gp.y; G(i, 6) = gp.z; G(i, 7) = gv.x; G(i, 8) = gv.y; G(i, 9) = gv.z;
        G(i, 10) = ga.x; G(i, 11) = ga.y; G(i, 12) = ga.z;
        // the root bone's own globals are replaced by the last frame's (test_fullframework.py:148-151)
        if (i == 0) { gr = Rr; gp = Rp; gv = Rv; ga = Ra; }
        const V3 xp = qrot(Ri, subv(gp, Rp));
        const Q xr = qmul(Ri, gr);
        const V3 xv = qrot(Ri, gv), xa = qrot(Ri, ga);
        // to_xform_xy, motion/quat.py:42-55
        const float x2 = xr.x + xr.x, y2 = xr.y + xr.y, z2 = xr.z + xr.z;
        const float xx = xr.x * x2, yy = xr.y * y2, wx = xr.w * x2;
        const float xy = xr.x * y2, yz = xr.y * z2, wy = xr.w * y2;
        const float xz = xr.x * z2, zz = xr.z * z2, wz = xr.w * z2;
        float* o = X + e * 15;
        o[0] = xp.x; o[1] = xp.y; o[2] = xp.z;
        o[3] = 1.0f - (yy + zz); o[4] = xy - wz;
        o[5] = xy + wz;          o[6] = 1.0f - (xx + zz);
        o[7] = xz - wy;          o[8] = yz + wx;
        o[9] = xv.x; o[10] = xv.y; o[11] = xv.z;
        o[12] = xa.x; o[13] = xa.y; o[14] = xa.z;
    }
}

hipError_t featurize_init() {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(&mocha_featurize), hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)(40 * FC * FT * sizeof(float)));
}

hipError_t launch_featurize(const float* Yrot, const float* Ypos, const float* Yvel, const float* Yang, const int* parents, float* X,
                            int B, int T, int J, hipStream_t s) {
    if (B <= 0) return hipSuccess;
    if (J > 40 || J < 1) return hipErrorInvalidValue;
    const int frames = B * T;
    hipLaunchKernelGGL(mocha_featurize, dim3((frames + FT - 1) / FT), dim3(FT), (size_t)J * FC * FT * sizeof(float), s, Yrot, Ypos, Yvel,
                       Yang, parents, X, frames, T, J);
    return hipGetLastError();
}

}  // namespace mocha
