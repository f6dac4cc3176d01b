// Bandwidth-bound kernels of the MOCHA path on gfx950: graph-adjacency front ends, instance
// norms, the final 64->15 projection, bank utilities.  All coalesced on the channel axis
// (activations are kept channel-last: rows = (window, time, node), columns = channels).
#include "kernels.h"
#include "device_utils.h"
#include <cstdlib>

namespace mocha {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float lrelu02(float x) { return x > 0.f ? x : 0.2f * x; }
// the same value for every x (max(x, 0.2 x): x for x >= 0, 0.2 x below) in two instructions; the instruction is written out because fmaxf
// brings a canonicalising v_max_f32 x, x per operand with it
__device__ __forceinline__ float lrelu02_max(float x) {
    const float y = 0.2f * x;
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y));
    return r;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// ---------------------------------------------------------------------------------------
// embed_front: model.py:43-46 up to (and including) the joint->body-part pool, which is linear
// and is commuted in front of the two convolutions of the joint ST-GCN block.
//   h[v][c]        = lrelu( sum_i X[v][i] W1[c][i] + b1[c] )                 (model.py:44, blocks.py:131)
//   out[pk][c]     = sum_v AP'[pk][v] h[v][c],  pk = 3 p + k,  AP'[3p+k][v] = (A_j[k] · Pool)[v][p]   (blocks.py:64, graph.py:463-465)
// Both contractions run on the fp32 matrix pipe, one frame per wave at a time (the scalar version spent 5x the HBM time in
// LDS-fed FMAs):
//   1. D1[joint][channel] = X_f (32 x 16, zero padded) · W1^T: 8 MFMA steps per 32-channel tile; X_f goes through a per-wave
//      LDS tile so that each lane's 8 features are two ds_read_b128;
//   2. the accumulator D1 (lane = channel, register s = joint (s&3) + 8 (s>>2) + 4 hh) IS the B operand of the second
//      contraction register by register, exactly as P^T in attention.hip: D2[pk][channel] = sum_s AP'[pk][joint(s, hh)] h[...]:
//      16 MFMA steps per tile with the AP' coefficients as per-lane constants.
// ---------------------------------------------------------------------------------------
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(256) void mocha_embed_front(const float* __restrict__ X, const float* __restrict__ W1,
                                                         const float* __restrict__ b1, const float* __restrict__ AP,
                                                         float* __restrict__ out, int nframes, int V, int Cin,
                                                         const float* __restrict__ xmean, const float* __restrict__ xstd, int raw_root) {
    // rows of 20 floats (16 + 4 pad): the per-lane ds_read_b128 of a joint's 8 features hit 16 distinct 16-byte units per 16-lane
    // service group (5 l mod 16 is a bijection); with 16-float rows they were 4-way conflicts (PMC: 0.59 conflict cycles per LDS cycle)
    __shared__ __attribute__((aligned(16))) float xs_all[4][32 * 20];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hh = lane >> 5;
    float* xs = xs_all[wave];

    // ---- per-lane constants
    float wb[2][8], bias[2], apv[16];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int c = t * 32 + l31;
        bias[t] = b1[c];
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const int feat = 2 * s + hh;
            wb[t][s] = feat < Cin ? W1[c * Cin + feat] : 0.f;
        }
    }
    {
        const int pk = l31, pp = pk / 3, kk = pk - pp * 3;
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const int joint = (s & 3) + 8 * (s >> 2) + 4 * hh;
            apv[s] = (pk < 18 && joint < V) ? AP[(kk * V + joint) * 6 + pp] : 0.f;
        }
    }
    // staging map of this lane's (at most 8) elements of a frame: element e = lane + 64 i -> joint e / Cin, feature e % Cin,
    // stored at joint * 20 + (feature & 1) * 8 + (feature >> 1) so that a lane's 8 features of one parity are contiguous
    const int nelem = V * Cin;
    int slot[8];
    float zm[8], zs[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int e = lane + 64 * i;
        const int v = e / Cin, ci = e - v * Cin;
        slot[i] = e < nelem ? v * 20 + (ci & 1) * 8 + (ci >> 1) : -1;
        zm[i] = (xmean && e < nelem) ? xmean[raw_root * Cin + e] : 0.f;
        zs[i] = (xmean && e < nelem) ? xstd[raw_root * Cin + e] : 1.f;
    }
    for (int i = lane; i < 32 * 20; i += 64) xs[i] = 0.f;          // padding joints / the padding feature stay zero
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();

    const int stride = gridDim.x * 4;
    int frame = blockIdx.x * 4 + wave;
    float xr[8];
    auto fetch = [&](int f) __attribute__((always_inline)) {
        // raw_root = 1: frames carry the root bone in front (V+1 joints), X = (X[:,:,1:] - mean) / std   (test_fullframework.py:186)
        const float* xf = X + (size_t)f * (V + raw_root) * Cin + raw_root * Cin;
#pragma unroll
        for (int i = 0; i < 8; ++i) xr[i] = slot[i] >= 0 ? xf[lane + 64 * i] : 0.f;
    };
    if (frame < nframes) fetch(frame);
    for (; frame < nframes; frame += stride) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
            if (slot[i] >= 0) xs[slot[i]] = xmean ? (xr[i] - zm[i]) / zs[i] : xr[i];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const f32x4 xa0 = *reinterpret_cast<const f32x4*>(xs + l31 * 20 + hh * 8);
        const f32x4 xa1 = *reinterpret_cast<const f32x4*>(xs + l31 * 20 + hh * 8 + 4);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (frame + stride < nframes) fetch(frame + stride);       // next frame's loads fly under this frame's MFMAs
        const float xa[8] = {xa0[0], xa0[1], xa0[2], xa0[3], xa1[0], xa1[1], xa1[2], xa1[3]};
        float* of = out + (size_t)frame * 18 * 64;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            f32x16 h;
#pragma unroll
            for (int r = 0; r < 16; ++r) h[r] = 0.f;
#pragma unroll
            for (int s = 0; s < 8; ++s) h = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[s], wb[t][s], h, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 16; ++r) h[r] = lrelu02(h[r] + bias[t]);
            f32x16 o;
#pragma unroll
            for (int r = 0; r < 16; ++r) o[r] = 0.f;
#pragma unroll
            for (int s = 0; s < 16; ++s) o = __builtin_amdgcn_mfma_f32_32x32x2f32(apv[s], h[s], o, 0, 0, 0);
            // o[r] = out[pk = (r&3) + 8 (r>>2) + 4 hh][channel t*32 + l31]; rows 18..31 are padding
#pragma unroll
            for (int r = 0; r < 10; ++r) {
                const int pk = (r & 3) + 8 * (r >> 2) + 4 * hh;
                if (pk < 18) of[pk * 64 + t * 32 + l31] = o[r];
            }
        }
    }
}

// ---------------------------------------------------------------------------------------
// embed_front on the bf16 matrix pipe: the same two chained contractions as plane products (gemm_x3.hip: every fp32 operand as three
// exact bf16 planes, six v_mfma_f32_32x32x16_bf16 passes per product, fp32 accumulate).  The first contraction's K (16 padded
// features) is ONE MFMA k step, the second's (32 padded joints) two: 36 MFMAs of 32 cycles per frame instead of 48 of 64.  W1 and AP'
// are per-lane plane constants; the frame's features are split after the LDS read; the accumulator h (lane = channel, registers =
// joints) is split in registers - registers 8j..8j+7 are, for lane half h, the joints 16j + 8(e>>2) + 4h + (e&3), one B operand
// of a K = 16 MFMA in that joint order (as P^T in attention_x3.hip), and AP' is held in the same order.
// ---------------------------------------------------------------------------------------
typedef short s16x8 __attribute__((ext_vector_type(8)));

struct EmbedConsts {
    s16x8 wpl[2][3], apl[2][3];       // W1 (this lane's channel of either half, its 8 features) and AP' (row pk = lane & 31) as planes
    float bias[2];
    int slot[8];                      // staging map of this lane's (at most 8) elements of a frame: e = lane + 64 i -> joint e / Cin, feature e % Cin
    float zm[8], zs[8];               // z-score of raw poses (mocha_set_pose_norm)
};
__device__ __forceinline__ void embed_consts(EmbedConsts& k, const float* __restrict__ W1, const float* __restrict__ b1,
                                             const float* __restrict__ AP, int V, int Cin, const float* __restrict__ xmean,
                                             const float* __restrict__ xstd, int raw_root, int lane) {
    const int l31 = lane & 31, hh = lane >> 5;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int c = t * 32 + l31;
        k.bias[t] = b1[c];
        float w8[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { const int feat = 8 * hh + e; w8[e] = feat < Cin ? W1[c * Cin + feat] : 0.f; }
        plane_split8(w8, k.wpl[t]);
    }
    {
        const int pk = l31, pp = pk / 3, kk = pk - pp * 3;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            float a8[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int joint = 16 * j + 8 * (e >> 2) + 4 * hh + (e & 3);
                a8[e] = (pk < 18 && joint < V) ? AP[(kk * V + joint) * 6 + pp] : 0.f;
            }
            plane_split8(a8, k.apl[j]);
        }
    }
    const int nelem = V * Cin;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int e = lane + 64 * i;
        const int v = e / Cin, ci = e - v * Cin;
        k.slot[i] = e < nelem ? v * 20 + ci : -1;
        k.zm[i] = (xmean && e < nelem) ? xmean[raw_root * Cin + e] : 0.f;
        k.zs[i] = (xmean && e < nelem) ? xstd[raw_root * Cin + e] : 1.f;
    }
}
// one frame: the staged (and z-scored) features go through this wave's LDS rows into MFMA operand order; both contractions.
// o[t][r] = out[pk = (r&3) + 8 (r>>2) + 4 hh][channel t*32 + l31]; rows 18..31 are padding
__device__ __forceinline__ void embed_stage_x3(const EmbedConsts& k, float* xs, const float (&xr)[8], bool zscore, float (&xa)[8], int lane) {
    const int l31 = lane & 31, hh = lane >> 5;
#pragma unroll
    for (int i = 0; i < 8; ++i)
        if (k.slot[i] >= 0) xs[k.slot[i]] = zscore ? (xr[i] - k.zm[i]) / k.zs[i] : xr[i];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const f32x4 xa0 = *reinterpret_cast<const f32x4*>(xs + l31 * 20 + hh * 8);
    const f32x4 xa1 = *reinterpret_cast<const f32x4*>(xs + l31 * 20 + hh * 8 + 4);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    xa[0] = xa0[0]; xa[1] = xa0[1]; xa[2] = xa0[2]; xa[3] = xa0[3]; xa[4] = xa1[0]; xa[5] = xa1[1]; xa[6] = xa1[2]; xa[7] = xa1[3];
}
// `narrow` (V <= 24, wave-uniform): the joints 24..31 of the second contraction are padding - AP' is zero there - so their half of the
// accumulator is neither activated nor split (a quarter of the frame's split instructions)
__device__ __forceinline__ void embed_frame_x3(const EmbedConsts& k, const float (&xa)[8], f32x16 (&o)[2], bool narrow) {
    s16x8 xpl[3];
    plane_split8(xa, xpl);
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        f32x16 h;                                                  // the bias is the accumulator's initial value
#pragma unroll
        for (int r = 0; r < 16; ++r) h[r] = k.bias[t];
#pragma unroll
        for (int pr = 0; pr < 6; ++pr) h = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xpl[PLANE_PA[pr]], k.wpl[t][PLANE_PB[pr]], h, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 16; ++r) o[t][r] = 0.f;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            s16x8 hpl[3];
            if (j == 1 && narrow) {
                const f32x4 h4 = {lrelu02_max(h[8]), lrelu02_max(h[9]), lrelu02_max(h[10]), lrelu02_max(h[11])};
                u32x2_t a[3];
                plane_split4(h4, a);
#pragma unroll
                for (int q = 0; q < 3; ++q) { const u32x4_t v = {a[q][0], a[q][1], 0u, 0u}; hpl[q] = __builtin_bit_cast(s16x8, v); }
            } else {
                float h8[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) h8[e] = lrelu02_max(h[8 * j + e]);
                plane_split8(h8, hpl);
            }
#pragma unroll
            for (int pr = 0; pr < 6; ++pr) o[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k.apl[j][PLANE_PA[pr]], hpl[PLANE_PB[pr]], o[t], 0, 0, 0);
        }
    }
}

__global__ __launch_bounds__(256) void mocha_embed_front_x3(const float* __restrict__ X, const float* __restrict__ W1,
                                                            const float* __restrict__ b1, const float* __restrict__ AP,
                                                            float* __restrict__ out, int nframes, int V, int Cin,
                                                            const float* __restrict__ xmean, const float* __restrict__ xstd, int raw_root) {
    __shared__ __attribute__((aligned(16))) float xs_all[4][32 * 20];       // rows of 20 floats: conflict-free b128 row reads
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hh = lane >> 5;
    float* xs = xs_all[wave];
    EmbedConsts k;
    embed_consts(k, W1, b1, AP, V, Cin, xmean, xstd, raw_root, lane);
    for (int i = lane; i < 32 * 20; i += 64) xs[i] = 0.f;          // padding joints / the padding feature stay zero
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();

    const int stride = gridDim.x * 4;
    int frame = blockIdx.x * 4 + wave;
    float xr[8];
    auto fetch = [&](int f) __attribute__((always_inline)) {
        const float* xf = X + (size_t)f * (V + raw_root) * Cin + raw_root * Cin;
#pragma unroll
        for (int i = 0; i < 8; ++i) xr[i] = k.slot[i] >= 0 ? xf[lane + 64 * i] : 0.f;
    };
    if (frame < nframes) fetch(frame);
    for (; frame < nframes; frame += stride) {
        float xa[8];
        embed_stage_x3(k, xs, xr, xmean != nullptr, xa, lane);
        if (frame + stride < nframes) fetch(frame + stride);       // next frame's loads fly under this frame's MFMAs
        f32x16 o[2];
        embed_frame_x3(k, xa, o, V <= 24);
        float* of = out + (size_t)frame * 18 * 64;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 10; ++r) {
                const int pk = (r & 3) + 8 * (r >> 2) + 4 * hh;
                if (pk < 18) of[pk * 64 + t * 32 + l31] = o[t][r];
            }
    }
}

// ---------------------------------------------------------------------------------------
// embed_sums: embed_front and window_sums<48> in one kernel (the folded joint block's operand, mocha_api.cpp: fold_joint) - the
// 192-channel frame rows never go to HBM (323 MB written and read again per demo step).  A workgroup's four waves take four
// consecutive frames per step into an LDS ring of 12 frames; after step s of a run of pooled frames t' = a .. a + n - 1 of one
// window (frames 4a - 2 + 4s + wave, reflected; n + 1 steps) the ring holds the 8 frames of t' = a + s - 1, whose five 4-frame sums the
// whole workgroup writes - the same additions in the same order as mocha_window_sums.  The pooled frames of all windows are cut into
// gridDim.x equal contiguous ranges (a range crosses windows; each piece of a window costs one extra step), one barrier per step.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void mocha_embed_sums_x3(const float* __restrict__ X, const float* __restrict__ W1,
                                                           const float* __restrict__ b1, const float* __restrict__ AP,
                                                           float* __restrict__ u, int nwin, int V, int Cin,
                                                           const float* __restrict__ xmean, const float* __restrict__ xstd, int raw_root) {
    __shared__ __attribute__((aligned(16))) float xs_all[4][32 * 20];
    __shared__ __attribute__((aligned(16))) float ring[12][18 * 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hh = lane >> 5;
    float* xs = xs_all[wave];
    EmbedConsts k;
    embed_consts(k, W1, b1, AP, V, Cin, xmean, xstd, raw_root, lane);
    for (int i = lane; i < 32 * 20; i += 64) xs[i] = 0.f;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();

    const long long T = (long long)nwin * 15;
    const int tp_begin = (int)(T * blockIdx.x / gridDim.x), tp_end = (int)(T * (blockIdx.x + 1) / gridDim.x);
    if (tp_begin >= tp_end) return;
    float xr[8];
    auto fetch = [&](int w, int a, int s) __attribute__((always_inline)) {
        int r = 4 * a - 2 + 4 * s + wave;
        r = r < 0 ? -r : r;
        r = r > 59 ? 118 - r : r;                                   // reflect padding (blocks.py:112-118)
        const float* xf = X + ((size_t)w * 60 + r) * (V + raw_root) * Cin + raw_root * Cin;
#pragma unroll
        for (int i = 0; i < 8; ++i) xr[i] = k.slot[i] >= 0 ? xf[lane + 64 * i] : 0.f;
    };
    int tp0 = tp_begin;
    int w = tp0 / 15, a = tp0 - w * 15, n = min(15 - a, tp_end - tp0), s = 0;
    fetch(w, a, 0);
    const f32x4* ring4 = reinterpret_cast<const f32x4*>(&ring[0][0]);
    for (bool more = true; more;) {
        float xa[8];
        embed_stage_x3(k, xs, xr, xmean != nullptr, xa, lane);
        // the step after this one: the next four frames of this run, or the first four of the next run
        int w2 = w, a2 = a, n2 = n, s2 = s + 1;
        if (s2 > n) {
            tp0 += n;
            if (tp0 >= tp_end) more = false;
            else { w2 = tp0 / 15; a2 = tp0 - w2 * 15; n2 = min(15 - a2, tp_end - tp0); s2 = 0; }
        }
        if (more) fetch(w2, a2, s2);
        f32x16 o[2];
        embed_frame_x3(k, xa, o, V <= 24);
        if (s == 0) __syncthreads();                                // the previous run's last sums are read before ring group 0 is rewritten
        float* of = ring[4 * (s % 3) + wave];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 10; ++r) {
                const int pk = (r & 3) + 8 * (r >> 2) + 4 * hh;
                if (pk < 18) of[pk * 64 + t * 32 + l31] = o[t][r];
            }
        __syncthreads();
        if (s >= 1) {
            // pooled frame a + s - 1: ring frames (relative) 4 (s - 1) .. 4 (s - 1) + 7; item = (body part, channel quad) = float4 index of a frame
            const int g0 = 4 * ((s - 1) % 3), g1 = 4 * (s % 3);
            f32x4* ur = reinterpret_cast<f32x4*>(u) + ((size_t)w * 15 + a + s - 1) * 6 * 240;
            {
                f32x4 f[8];
#pragma unroll
                for (int i = 0; i < 4; ++i) { f[i] = ring4[(g0 + i) * 288 + tid]; f[4 + i] = ring4[(g1 + i) * 288 + tid]; }
                const int part = tid / 48, q = tid - part * 48;
#pragma unroll
                for (int dt = 0; dt < 5; ++dt) ur[part * 240 + dt * 48 + q] = (((f[dt] + f[dt + 1]) + f[dt + 2]) + f[dt + 3]) * 0.25f;
            }
            if (tid < 160) {                                        // the last 32 items, one (item, tap) pair per thread
                const int it = 256 + tid / 5, dt = tid - (tid / 5) * 5;
                f32x4 f[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) { const int rel = dt + i; f[i] = ring4[((rel < 4 ? g0 : g1) + (rel & 3)) * 288 + it]; }
                const int part = it / 48, q = it - part * 48;
                ur[part * 240 + dt * 48 + q] = (((f[0] + f[1]) + f[2]) + f[3]) * 0.25f;
            }
        }
        w = w2; a = a2; n = n2; s = s2;
    }
}
// Workgroups of the plane build: every workgroup pays the per-lane plane constants (W1, AP', the staging map) before its first frame, so the
// grid is what the chip holds at once (2 workgroups of 4 waves per CU at 158 + 32 registers) and each wave takes ~17 frames of the demo
// step: 78 -> 65 us per launch (profiles/r04/h_embed_front_ab.txt; 256: 79, 1024: 69, uncapped: 122).  Option "embed_front_max_wgs" (per
// context, passed in as max_wgs; default 512).

hipError_t launch_embed_front(const float* X, const float* W1, const float* b1, const float* AP, float* out,
                              int nframes, int V, int Cin, const float* xmean, const float* xstd, int raw_root, hipStream_t s, bool planes,
                              int max_wgs) {
    if (nframes <= 0) return hipSuccess;
    if (V > 32 || Cin > 16 || V * Cin > 512) return hipErrorInvalidValue;
    const int wgs = (nframes + 3) / 4;
    const int cap = !planes ? 2048 : max_wgs > 0 ? max_wgs : 512;
    if (planes)
        hipLaunchKernelGGL(mocha_embed_front_x3, dim3(wgs < cap ? wgs : cap), dim3(256), 0, s, X, W1, b1, AP, out,
                           nframes, V, Cin, xmean, xstd, raw_root);
    else
        hipLaunchKernelGGL(mocha_embed_front, dim3(wgs < 2048 ? wgs : 2048), dim3(256), 0, s, X, W1, b1, AP, out,
                           nframes, V, Cin, xmean, xstd, raw_root);
    return hipGetLastError();
}

hipError_t launch_embed_sums(const float* X, const float* W1, const float* b1, const float* AP, float* u, int nwin, int V, int Cin,
                             const float* xmean, const float* xstd, int raw_root, hipStream_t s, int max_wgs) {
    if (nwin <= 0) return hipSuccess;
    if (V > 32 || Cin > 16 || V * Cin > 512) return hipErrorInvalidValue;
    const long long T = (long long)nwin * 15;
    const int cap = max_wgs > 0 ? max_wgs : 512;
    hipLaunchKernelGGL(mocha_embed_sums_x3, dim3((unsigned)(T < cap ? T : cap)), dim3(256), 0, s, X, W1, b1, AP, u, nwin, V, Cin, xmean, xstd, raw_root);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------
// body_front: LeakyReLU then the body-part adjacency (commuted in front of the 1x1 conv):
//   out[(f,w)][k*256+c] = sum_v A_b[k][v][w] lrelu(x[(f,v)][c])      (blocks.py:131, :64)
// one thread per (frame f, 4 channels); the 72 coefficients are wave-uniform scalar loads.
// No packed fp32 instructions here: the build whose compiler-formed v_pk_fma_f32 took the HIGH register of a coefficient pair for
// the low result (op_sel) intermittently produced 0 there, lanes 48-63, whenever workgroups of another stream's plane GEMM - bf16
// MFMAs with ordinary VALU instructions issued between them - shared the CU (tools/body_front_repro.hip reproduces it standalone;
// tools/experiments/README.md, "two streams").  The coefficients come from scalar loads, so no LDS and no barrier either.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) MOCHA_NO_PACKED_F32
void mocha_body_front(const float* __restrict__ x, const float* __restrict__ Ab, float* __restrict__ out, int frames) {
    const int gid = blockIdx.x * 256 + threadIdx.x;
    const int f = gid >> 6, c4 = (gid & 63) * 4;
    if (f >= frames) return;
    f32x4 xv[6];
#pragma unroll
    for (int v = 0; v < 6; ++v) {
        f32x4 t = *reinterpret_cast<const f32x4*>(x + ((size_t)f * 6 + v) * 256 + c4);
        t[0] = lrelu02(t[0]); t[1] = lrelu02(t[1]); t[2] = lrelu02(t[2]); t[3] = lrelu02(t[3]);
        xv[v] = t;
    }
#pragma unroll
    for (int w = 0; w < 6; ++w)
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int v = 0; v < 6; ++v) acc += xv[v] * Ab[(k * 6 + v) * 6 + w];
            *reinterpret_cast<f32x4*>(out + ((size_t)f * 6 + w) * 512 + k * 256 + c4) = acc;
        }
}

hipError_t launch_body_front(const float* x, const float* A_b, float* out, int rows6, hipStream_t s) {
    if (rows6 <= 0) return hipSuccess;
    const long long threads = (long long)rows6 * 64;
    hipLaunchKernelGGL(mocha_body_front, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, x, A_b, out, rows6);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------
// joint_expand: body-part -> joint copy (graph.py:606-608) followed by the joint adjacency of the
// output ST-GCN block (blocks.py:64) on the 3x64 gcn channels, at the 15-frame resolution (the
// nearest x4 upsampling, model.py:74, repeats frames and is folded into the next GEMM's gather):
//   out[(f,w)][c] = sum_k sum_p AU[k][p][w] g[(f,p)][k*64+c],   AU[k][p][w] = sum_{v in part p} A_j[k][v][w]
// ---------------------------------------------------------------------------------------
// Round 4: a wave owns a frame and a lane a channel - the 18 inputs of that channel (6 parts x 3 gcn slices) are fetched straight into
// registers (coalesced 256-byte runs); per input the V coefficients of its row of AU are wave-uniform and contiguous (scalar loads), one
// accumulator per joint; no LDS, no barrier.  The first build staged a frame in LDS and read two LDS words per FMA: LDS-bound at 0.33 of
// the HBM peak.  Same products in the same order per output.  VT = joints at compile time (22, 24) or 0: any V <= 32, a loop over joints.
template <int VT>
__global__ __launch_bounds__(256) void mocha_joint_expand(const float* __restrict__ g, const float* __restrict__ AU,
                                                          float* __restrict__ out, int frames, int V) {
    const int c = threadIdx.x & 63;
    const int f = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (f >= frames) return;
    const float* gf = g + (size_t)f * 6 * 192 + c;
    float gv[3][6];
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int p = 0; p < 6; ++p) gv[k][p] = gf[p * 192 + k * 64];
    if (VT > 0) {
        float* of = out + (size_t)f * VT * 64 + c;
        float acc[VT > 0 ? VT : 1];
#pragma unroll
        for (int w = 0; w < VT; ++w) acc[w] = 0.f;
#pragma unroll
        for (int k = 0; k < 3; ++k)
#pragma unroll
            for (int p = 0; p < 6; ++p)
#pragma unroll
                for (int w = 0; w < VT; ++w) acc[w] = fmaf(AU[(k * 6 + p) * VT + w], gv[k][p], acc[w]);
#pragma unroll
        for (int w = 0; w < VT; ++w) of[w * 64] = acc[w];
    } else {
        float* of = out + (size_t)f * V * 64 + c;
        for (int w = 0; w < V; ++w) {
            float a = 0.f;
#pragma unroll
            for (int k = 0; k < 3; ++k)
#pragma unroll
                for (int p = 0; p < 6; ++p) a = fmaf(AU[(k * 6 + p) * V + w], gv[k][p], a);
            of[w * 64] = a;
        }
    }
}

hipError_t launch_joint_expand(const float* g, const float* AU, float* out, int nframes15, int V, hipStream_t s) {
    if (nframes15 <= 0) return hipSuccess;
    if (V > 32) return hipErrorInvalidValue;
    const dim3 grid((nframes15 + 3) / 4);
    if (V == 22) hipLaunchKernelGGL(mocha_joint_expand<22>, grid, dim3(256), 0, s, g, AU, out, nframes15, V);
    else if (V == 24) hipLaunchKernelGGL(mocha_joint_expand<24>, grid, dim3(256), 0, s, g, AU, out, nframes15, V);
    else hipLaunchKernelGGL(mocha_joint_expand<0>, grid, dim3(256), 0, s, g, AU, out, nframes15, V);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------
// final_proj: Y[m][o] = sum_c lrelu(z[m][c]) W6[o][c] + b6[o]     (model.py:77-79), optionally de-normalised.
// On the fp32 matrix pipe: a wave owns 32 rows; lane (row, k-half) fetches its row's 8 float4 straight into MFMA operand
// layout (the k permutation is shared with the per-lane W6 constants), 32 MFMA steps give D[row][o] with the lane on o;
// the 128 x 15 results of a workgroup go through LDS so that the Y rows (60 bytes each) leave as one contiguous run.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void mocha_final_proj(const float* __restrict__ z, const float* __restrict__ W6,
                                                        const float* __restrict__ b6, float* __restrict__ Y, int rows,
                                                        int Cout, int V, const float* __restrict__ ymean,
                                                        const float* __restrict__ ystd, int phased) {
    __shared__ float ys[128 * 16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hh = lane >> 5;
    const int r0 = blockIdx.x * 128;
    const int row = r0 + wave * 32 + l31;
    // phased (mocha_api.cpp, fold_upsample): z holds rows (window, source frame s, joint) x (phase, channel); output row (window, t, joint)
    // reads the 64 channels of phase t & 3 in row (window, t >> 2, joint)
    const float* zr = z + (size_t)row * 64;
    if (phased && row < rows) {
        const int v = row % V, bt = row / V, t = bt % 60, b = bt / 60;
        zr = z + ((size_t)(b * 15 + (t >> 2)) * V + v) * 256 + (t & 3) * 64;
    }
    f32x4 a[8], w[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        f32x4 zero = {0.f, 0.f, 0.f, 0.f};
        a[q] = row < rows ? *reinterpret_cast<const f32x4*>(zr + 8 * q + 4 * hh) : zero;
        w[q] = l31 < Cout ? *reinterpret_cast<const f32x4*>(W6 + l31 * 64 + 8 * q + 4 * hh) : zero;
    }
    f32x16 d;
#pragma unroll
    for (int r = 0; r < 16; ++r) d[r] = 0.f;
#pragma unroll
    for (int q = 0; q < 8; ++q)
#pragma unroll
        for (int e = 0; e < 4; ++e) d = __builtin_amdgcn_mfma_f32_32x32x2f32(lrelu02(a[q][e]), w[q][e], d, 0, 0, 0);
    // d[r] = sum for (row = wave*32 + (r&3) + 8 (r>>2) + 4 hh, o = l31)
    if (l31 < Cout) {
        const float bo = b6[l31];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int rl = wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
            float y = d[r] + bo;
            if (ymean) {   // de-normalise, Y * Y_std[0,:,1:] + Y_mean[0,:,1:] (test_fullframework.py:303,457); norms carry the root row
                const int v = (r0 + rl) % V;
                y = y * ystd[(v + 1) * Cout + l31] + ymean[(v + 1) * Cout + l31];
            }
            ys[rl * Cout + l31] = y;
        }
    }
    __syncthreads();
    const int nrow = (rows - r0) < 128 ? (rows - r0) : 128;
    for (int i = tid; i < nrow * Cout; i += 256) Y[(size_t)r0 * Cout + i] = ys[i];
}

hipError_t launch_final_proj(const float* z, const float* W6, const float* b6, float* Y, int rows, int Cout, int V,
                             const float* ymean, const float* ystd, hipStream_t s, int phased) {
    if (rows <= 0) return hipSuccess;
    if (Cout > 16) return hipErrorInvalidValue;
    hipLaunchKernelGGL(mocha_final_proj, dim3((rows + 127) / 128), dim3(256), 0, s, z, W6, b6, Y, rows, Cout, V, ymean, ystd, phased);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------
// instance norm over the token axis, per (window, channel)  (net/transformer.py:13-20):
//   mean = sum/n ; std = sqrt(sum (x-mean)^2 / (n-1)) ; out = (x-mean)/(std+1e-5)
// One workgroup per window.  Thread = (channel quad q = tid&63, token group g = tid>>6): it keeps
// its <= 24 tokens x 4 channels in registers (one 16-byte load each), so the window is read once;
// the four token groups combine their partial sums through LDS.
// ---------------------------------------------------------------------------------------
// 8 token groups (512 threads) per window: 12 tokens x 4 channels per thread keeps the kernels under 100 VGPRs (the
// 4-group version held 24 tokens per thread: 165-224 VGPRs, 2-3 waves per SIMD for a bandwidth-bound kernel).
static constexpr int IN_NG = 8;             // token groups per window
static constexpr int IN_MAXT = 12;          // tokens per thread (n <= 96)

// QW = channel quads per workgroup: 16 (a window's 256 channels over four workgroups; the default for every batch since round 4) or 64 (one
// workgroup per window).  A channel's tokens are
// partitioned and summed in the same order either way: results are bit-identical.
template <int QW>
__device__ __forceinline__ f32x4 group_sum4(f32x4 v, f32x4* red, int q, int g) {
    __syncthreads();                         // red[] may still be read from the previous reduction
    red[g * QW + q] = v;
    __syncthreads();
    f32x4 a = red[q];
#pragma unroll
    for (int k = 1; k < IN_NG; ++k) a += red[k * QW + q];      // fixed order: deterministic
    return a;
}

template <int QW>
__device__ __forceinline__ void inorm_stats(const f32x4* xv, int cnt, int n, f32x4* red, int q, int g, f32x4& mean, f32x4& den, f32x4* sd = nullptr) {
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < IN_MAXT; ++i)
        if (i < cnt) s += xv[i];
    mean = group_sum4<QW>(s, red, q, g) / (float)n;
    f32x4 qq = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < IN_MAXT; ++i)
        if (i < cnt) { const f32x4 d = xv[i] - mean; qq += d * d; }
    qq = group_sum4<QW>(qq, red, q, g) / (float)(n - 1);
    const f32x4 sq = {sqrtf(qq[0]), sqrtf(qq[1]), sqrtf(qq[2]), sqrtf(qq[3])};
    den = sq + 1e-5f;
    if (sd) *sd = sq;                                        // the std itself: den - eps loses it when s << eps
}

template <int QW>
__global__ __launch_bounds__(QW * IN_NG) void mocha_instnorm(const float* __restrict__ x, float* __restrict__ out,
                                                             float* __restrict__ mean_out, const float* __restrict__ gm,
                                                             const float* __restrict__ gs, float* __restrict__ zn, int n, InormExtra ex) {
    __shared__ f32x4 red[QW * IN_NG];
    const int b = blockIdx.x, ql = threadIdx.x % QW, g = threadIdx.x / QW;
    const int q = blockIdx.y * QW + ql;                     // channel quad 0..63
    const int cnt = (n - g + IN_NG - 1) / IN_NG;           // tokens g, g + NG, ...
    const float* xrow = x + (size_t)b * n * 256;
    if (ex.row_idx) {                                       // gathered input: cha_encoded[frame_index] (test_fullframework.py:298, 465), index clamped
        long long r = ex.row_idx[b];
        r = r < 0 ? 0 : (r >= ex.table_rows ? ex.table_rows - 1 : r);
        xrow = ex.table + (size_t)r * n * 256;
    }
    const f32x4* xb = reinterpret_cast<const f32x4*>(xrow) + q;
    f32x4 xv[IN_MAXT];
#pragma unroll
    for (int i = 0; i < IN_MAXT; ++i)
        if (i < cnt) xv[i] = xb[(size_t)(g + IN_NG * i) * 64];
    if (ex.copy_out) {
        f32x4* cb = reinterpret_cast<f32x4*>(ex.copy_out + (size_t)b * n * 256) + q;
#pragma unroll
        for (int i = 0; i < IN_MAXT; ++i)
            if (i < cnt) cb[(size_t)(g + IN_NG * i) * 64] = xv[i];
    }
    // the z-score's operands do not depend on the statistics: the small-batch variant (latency-bound) fetches them now, under the
    // two reductions; the large-batch one (bandwidth-bound, register-lean for occupancy) where they are used
    constexpr bool EARLY = QW < 64;
    f32x4 zm[EARLY ? IN_MAXT : 1], zs[EARLY ? IN_MAXT : 1];
    if (EARLY && zn) {
#pragma unroll
        for (int i = 0; i < IN_MAXT; ++i)
            if (i < cnt) {
                const int t = g + IN_NG * i;
                zm[EARLY ? i : 0] = reinterpret_cast<const f32x4*>(gm)[t * 64 + q]; zs[EARLY ? i : 0] = reinterpret_cast<const f32x4*>(gs)[t * 64 + q];
            }
    }
    f32x4 mean, den;
    inorm_stats<QW>(xv, cnt, n, red, ql, g, mean, den);
    if (mean_out && g == 0) reinterpret_cast<f32x4*>(mean_out + (size_t)b * 256)[q] = mean;
    if (ex.mean64) {
        // the style MLP's input (AdaptiveAvgPool1d over the tokens, net/transformer.py:100-101) summed in float64: the MLP that follows runs
        // in float64 too (mocha_linear_f64), so that gamma / beta carry no rounding of their own into AdaIN (tools/precision_study.py)
        __shared__ double red64[QW * IN_NG][4];
        double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
#pragma unroll
        for (int i = 0; i < IN_MAXT; ++i)
            if (i < cnt) { a0 += (double)xv[i][0]; a1 += (double)xv[i][1]; a2 += (double)xv[i][2]; a3 += (double)xv[i][3]; }
        red64[g * QW + ql][0] = a0; red64[g * QW + ql][1] = a1; red64[g * QW + ql][2] = a2; red64[g * QW + ql][3] = a3;
        __syncthreads();
        if (g < 4) {                                         // token group g sums channel g of the quad over the groups, fixed order
            double a = red64[ql][g];
#pragma unroll
            for (int k = 1; k < IN_NG; ++k) a += red64[k * QW + ql][g];
            ex.mean64[(size_t)b * 256 + 4 * q + g] = a / (double)n;
        }
    }
    f32x4* ob = reinterpret_cast<f32x4*>(out + (size_t)b * n * 256) + q;
    float qs_n = 0.f, qs_d = 0.f;                            // this thread's share of ||zc||^2 and of the planes' residual (InormExtra::qstat)
#pragma unroll
    for (int i = 0; i < IN_MAXT; ++i)
        if (i < cnt) {
            const int t = g + IN_NG * i;
            const f32x4 v = (xv[i] - mean) / den;
            if (out) ob[(size_t)t * 64] = v;                   // out == nullptr: only the z-scored copy is wanted (characterize: cnt itself is not an output)
            if (ex.kvimg) {
                // the decoder attention's key / value images (attention_kv.hip): this thread's four channels of token t, split into bf16
                // planes - K from the normalised values, V from the input - 8 bytes per plane; eight neighbouring lanes fill a row's 64 bytes
                unsigned char* img = reinterpret_cast<unsigned char*>(ex.kvimg) + (size_t)b * ATTN_KV_IMG_BYTES + (q >> 3) * ATTN_KV_STAGE_BYTES + t * 64;
                u32x2_t pk[3], pv[3];
                plane_split4(v, pk);
                plane_split4(xv[i], pv);
                const int kpos = ((((q & 7) >> 1) ^ ((t >> 2) & 3)) << 4) + ((q & 1) << 3);
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) {
                    *reinterpret_cast<u32x2_t*>(img + pl * (ATTN_KV_STAGE_BYTES / 3) + kpos) = pk[pl];
                    *reinterpret_cast<u32x2_t*>(img + 8 * ATTN_KV_STAGE_BYTES + pl * (ATTN_KV_STAGE_BYTES / 3) + (q & 7) * 8) = pv[pl];
                }
            }
            if (zn) {
                const f32x4 m = EARLY ? zm[EARLY ? i : 0] : reinterpret_cast<const f32x4*>(gm)[t * 64 + q];
                const f32x4 sd = EARLY ? zs[EARLY ? i : 0] : reinterpret_cast<const f32x4*>(gs)[t * 64 + q];
                const f32x4 z = (v - m) / sd;
                (reinterpret_cast<f32x4*>(zn + (size_t)b * n * 256) + q)[(size_t)t * 64] = z;
                if (ex.zc || ex.zc16) {
                    const f32x4 zc = z - reinterpret_cast<const f32x4*>(ex.centre)[t * 64 + q];
                    if (ex.zc) (reinterpret_cast<f32x4*>(ex.zc + (size_t)b * n * 256) + q)[(size_t)t * 64] = zc;
                    f32x4 rest = {0.f, 0.f, 0.f, 0.f};
                    if (ex.zc16) {
                        const u32x2_t w = {bf16_bits(zc[0]) | (bf16_bits(zc[1]) << 16), bf16_bits(zc[2]) | (bf16_bits(zc[3]) << 16)};
                        (reinterpret_cast<u32x2_t*>(ex.zc16 + (size_t)b * n * 256) + q)[(size_t)t * 64] = w;
                        rest = zc - f32x4{__uint_as_float(w[0] << 16), __uint_as_float(w[0] & 0xffff0000u), __uint_as_float(w[1] << 16), __uint_as_float(w[1] & 0xffff0000u)};
                        if (ex.plane_stride > 0) {
                            const u32x2_t w1 = {bf16_bits(rest[0]) | (bf16_bits(rest[1]) << 16), bf16_bits(rest[2]) | (bf16_bits(rest[3]) << 16)};
                            (reinterpret_cast<u32x2_t*>(ex.zc16 + ex.plane_stride + (size_t)b * n * 256) + q)[(size_t)t * 64] = w1;
                            rest -= f32x4{__uint_as_float(w1[0] << 16), __uint_as_float(w1[0] & 0xffff0000u), __uint_as_float(w1[1] << 16), __uint_as_float(w1[1] & 0xffff0000u)};
                        }
                    }
                    if (ex.qstat) {
#pragma unroll
                        for (int k = 0; k < 4; ++k) { qs_n = fmaf(zc[k], zc[k], qs_n); qs_d = fmaf(rest[k], rest[k], qs_d); }
                    }
                }
            }
        }
    if (ex.qstat) {                                          // this workgroup's share of the row statistics: wave sums, then the waves in order;
        __shared__ float qsr[2][QW * IN_NG / 64 > 0 ? QW * IN_NG / 64 : 1];      // part blockIdx.y of QSTAT_PARTS (the consumer adds the parts in order)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { qs_n += __shfl_xor(qs_n, o); qs_d += __shfl_xor(qs_d, o); }
        if ((threadIdx.x & 63) == 0) { qsr[0][threadIdx.x >> 6] = qs_n; qsr[1][threadIdx.x >> 6] = qs_d; }
        __syncthreads();
        if (threadIdx.x == 0) {
            float a = 0.f, d = 0.f;
            for (int w = 0; w < QW * IN_NG / 64; ++w) { a += qsr[0][w]; d += qsr[1][w]; }
            float* qo = ex.qstat + (size_t)b * 2 * QSTAT_PARTS;
            qo[2 * blockIdx.y] = a; qo[2 * blockIdx.y + 1] = d;
            if (gridDim.y == 1)
                for (int k = 1; k < QSTAT_PARTS; ++k) { qo[2 * k] = 0.f; qo[2 * k + 1] = 0.f; }
        }
    }
    if (ex.kvimg && g < 96 - n) {                            // rows n .. 95 of both images: zero (a padded key's score is masked, its value row multiplies P = 0)
        unsigned char* img = reinterpret_cast<unsigned char*>(ex.kvimg) + (size_t)b * ATTN_KV_IMG_BYTES + (q >> 3) * ATTN_KV_STAGE_BYTES + (n + g) * 64 + (q & 7) * 8;
        const u32x2_t z = {0u, 0u};
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
            *reinterpret_cast<u32x2_t*>(img + pl * (ATTN_KV_STAGE_BYTES / 3)) = z;
            *reinterpret_cast<u32x2_t*>(img + 8 * ATTN_KV_STAGE_BYTES + pl * (ATTN_KV_STAGE_BYTES / 3)) = z;
        }
    }
}

// a handful of windows: four workgroups per window
// Windows up to which a window's 256 channels go over four workgroups (option "inorm_split_max").  Round 4: EVERY batch - the variant built
// for a handful of windows is also the faster one at 585 / 1 170 (mvn 77 -> 59 us, adain 43 -> 40, in_cha 43 -> 40; eight workgroups
// per window: no further gain; tools/ab/inorm_ab.py): four times the workgroups overlap each other's load -> reduce -> store phases.
// (per context since round 5: InormExtra::split_max / launch_adain's split_max; default: every batch)

hipError_t launch_instnorm(const float* x, float* out, float* mean_out, const float* gm, const float* gs, float* zn,
                           int B, int n, hipStream_t s, const InormExtra* exp) {
    if (B <= 0) return hipSuccess;
    if (n > IN_NG * IN_MAXT || n < 2) return hipErrorInvalidValue;
    InormExtra ex = exp ? *exp : InormExtra{};
    if (((ex.zc || ex.zc16) && (!zn || !ex.centre)) || (ex.row_idx && (!ex.table || ex.table_rows < 1))) return hipErrorInvalidValue;
    if (ex.kvimg && n < 96 - IN_NG) return hipErrorInvalidValue;      // the image's zero rows n .. 95 are written by the first 96 - n token groups
    if (ex.qstat && !(ex.zc || ex.zc16)) return hipErrorInvalidValue;
    if ((long long)ex.plane_stride < 0 || (ex.plane_stride > 0 && !ex.zc16)) return hipErrorInvalidValue;
    static_assert(QSTAT_PARTS == 4, "the four workgroups of a window write one part of the row statistics each");
    if (B <= ex.split_max) hipLaunchKernelGGL(mocha_instnorm<16>, dim3(B, 4), dim3(16 * IN_NG), 0, s, x, out, mean_out, gm, gs, zn, n, ex);
    else hipLaunchKernelGGL(mocha_instnorm<64>, dim3(B, 1), dim3(64 * IN_NG), 0, s, x, out, mean_out, gm, gs, zn, n, ex);
    return hipGetLastError();
}

// AdaIN followed by the attention's own mapping norm (net/transformer.py:108-113 then :49-56):
//   xad = (1+gamma) * IN(x) + beta ;  qin = IN(xad)
// closed != 0 (default, option "adain_closed_form"): qin from the FIRST statistics.  With m, s the mean / unbiased std of x over the
// tokens, xad's own token mean is exactly beta and its std |1+gamma| s / (s+eps), hence
//   qin = IN(xad) = (1+gamma) (x-m) / (|1+gamma| s + eps (s+eps))
// The literal order subtracts mean(xad) ~ beta from values whose spread is |1+gamma|: wherever a channel's 1+gamma is small the
// reference's own fp32 evaluation loses that channel's digits to the cancellation (tools/precision_study.py: two fp32 runs of the
// reference differ by up to 2e-3 on such inputs); the closed form has no cancellation and one reduction fewer.  closed == 0 keeps
// the literal two-pass order for A/B (tools/structured_matrix.py).
template <int QW>
__global__ __launch_bounds__(QW * IN_NG) void mocha_adain(const float* __restrict__ x, const float* __restrict__ gb, int gb_stride,
                                                          float* __restrict__ xad, float* __restrict__ qin, int n, int closed,
                                                          const int32_t* __restrict__ gb_idx, long long gb_rows) {
    __shared__ f32x4 red[QW * IN_NG];
    const int b = blockIdx.x, ql = threadIdx.x % QW, g = threadIdx.x / QW;
    const int q = blockIdx.y * QW + ql;
    const int cnt = (n - g + IN_NG - 1) / IN_NG;
    const f32x4* xb = reinterpret_cast<const f32x4*>(x + (size_t)b * n * 256) + q;
    f32x4 xv[IN_MAXT];
#pragma unroll
    for (int i = 0; i < IN_MAXT; ++i)
        if (i < cnt) xv[i] = xb[(size_t)(g + IN_NG * i) * 64];
    long long gr = b;
    if (gb_idx) { gr = gb_idx[b]; gr = gr < 0 ? 0 : (gr >= gb_rows ? gb_rows - 1 : gr); }      // the matched bank entry's constants, index clamped
    f32x4 gamma1 = reinterpret_cast<const f32x4*>(gb + (size_t)gr * gb_stride)[q];
    const f32x4 beta = reinterpret_cast<const f32x4*>(gb + (size_t)gr * gb_stride + 256)[q];
    gamma1 += 1.f;
    f32x4 mean, den, sd;
    inorm_stats<QW>(xv, cnt, n, red, ql, g, mean, den, &sd);
    f32x4* ab = reinterpret_cast<f32x4*>(xad + (size_t)b * n * 256) + q;
    f32x4* qb = reinterpret_cast<f32x4*>(qin + (size_t)b * n * 256) + q;
    if (closed) {
        f32x4 a1, a2;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            a1[k] = gamma1[k] / den[k];
            a2[k] = gamma1[k] / fmaf(fabsf(gamma1[k]), sd[k], 1e-5f * den[k]);
        }
#pragma unroll
        for (int i = 0; i < IN_MAXT; ++i)
            if (i < cnt) {
                const f32x4 d = xv[i] - mean;
                ab[(size_t)(g + IN_NG * i) * 64] = a1 * d + beta;
                qb[(size_t)(g + IN_NG * i) * 64] = a2 * d;
            }
        return;
    }
#pragma unroll
    for (int i = 0; i < IN_MAXT; ++i)
        if (i < cnt) {
            xv[i] = gamma1 * ((xv[i] - mean) / den) + beta;
            ab[(size_t)(g + IN_NG * i) * 64] = xv[i];
        }
    inorm_stats<QW>(xv, cnt, n, red, ql, g, mean, den);
#pragma unroll
    for (int i = 0; i < IN_MAXT; ++i)
        if (i < cnt) qb[(size_t)(g + IN_NG * i) * 64] = (xv[i] - mean) / den;
}

hipError_t launch_adain(const float* x, const float* gb, int gb_stride, float* xad, float* qin, int B, int n, hipStream_t s, int closed,
                        const int32_t* gb_idx, long long gb_rows, int split_max) {
    if (B <= 0) return hipSuccess;
    if (n > IN_NG * IN_MAXT || n < 2 || gb_stride < 512 || (gb_stride & 3) || (gb_idx && gb_rows < 1)) return hipErrorInvalidValue;
    if (B <= split_max) hipLaunchKernelGGL(mocha_adain<16>, dim3(B, 4), dim3(16 * IN_NG), 0, s, x, gb, gb_stride, xad, qin, n, closed, gb_idx, gb_rows);
    else hipLaunchKernelGGL(mocha_adain<64>, dim3(B, 1), dim3(64 * IN_NG), 0, s, x, gb, gb_stride, xad, qin, n, closed, gb_idx, gb_rows);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------
// Y = act(X W^T + bias) in float64 (the decoder's style MLP, net/transformer.py:102-107): X (M, ldx) float64, W (N, K) float64
// (the fp32 weights converted once), float64 accumulation; the result is stored as float64 (y64) and / or rounded once to fp32 (y32).
// gridDim.z = L independent blocks: block l reads X columns [l*xcol, l*xcol + K), W + l*N*K, bias + l*N and writes columns [l*N, (l+1)*N).
// Why float64: AdaIN's (1 + gamma) multiplies the normalised activations and the attention's mapping norm divides by |1 + gamma| s + eps:
// where 1 + gamma is within ~1e-4 of zero the fp32 rounding of gamma (1e-6 absolute) moves that channel of the queries by per cent
// - the largest single contribution to the reference's own fp32 error on structured inputs (tools/precision_study.py, part 2).
// 64 x 64 (or 32 x 32) tile per 256-thread workgroup, K staged through LDS 32 deep with the next tile's operands in registers.
// ---------------------------------------------------------------------------------------
// RT x RT outputs per thread: RT = 4 is the 64 x 64 tile, RT = 2 a 32 x 32 tile - four times the workgroups for launches that would not
// put one 64 x 64 tile on every CU (585 windows x 1024 columns: 160 tiles; the kernel is a chain of short K tiles with two barriers each,
// so co-resident workgroups, not the tile's arithmetic intensity, set its time)
template <int ACT /* 0 none, 2 LeakyReLU(0.2) */, int RT>
__global__ __launch_bounds__(256) void mocha_linear_f64(const double* __restrict__ X, int ldx, int xcol, const double* __restrict__ W,
                                                        const double* __restrict__ bias, double* __restrict__ y64, float* __restrict__ y32,
                                                        int ldy, int M, int N, int K) {
    constexpr int KT = 32, TILE = 16 * RT, LDT = TILE + 2;
    constexpr int KPT = 2 * RT;                                           // consecutive k a loader thread takes: TILE rows x 32 k over 256 threads
    __shared__ double Xs[KT][LDT];
    __shared__ double Ws[KT][LDT];
    const int l = blockIdx.z;
    const int r0 = blockIdx.y * TILE, n0 = blockIdx.x * TILE;
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const double* Xl = X + (size_t)l * xcol;
    const double* Wl = W + (size_t)l * N * K;
    double acc[RT][RT] = {};
    const int lr = threadIdx.x / (KT / KPT), lk = (threadIdx.x % (KT / KPT)) * KPT;      // loader: row / column lr of the tile, KPT consecutive k
    // rows / columns past the edge read the last valid one (their products are never stored): no predicated loads in the loop
    typedef double f64x2 __attribute__((ext_vector_type(2)));
    const int xrow = (r0 + lr) < M ? (r0 + lr) : M - 1, wrow = (n0 + lr) < N ? (n0 + lr) : N - 1;
    const f64x2* xp = reinterpret_cast<const f64x2*>(Xl + (size_t)xrow * ldx + lk);
    const f64x2* wp = reinterpret_cast<const f64x2*>(Wl + (size_t)wrow * K + lk);
    f64x2 xr[RT], wr[RT];
#pragma unroll
    for (int i = 0; i < RT; ++i) { xr[i] = xp[i]; wr[i] = wp[i]; }
    for (int k0 = 0; k0 < K; k0 += KT) {
        __syncthreads();                                                  // the previous tile has been read
#pragma unroll
        for (int i = 0; i < RT; ++i) {
            Xs[lk + 2 * i][lr] = xr[i][0]; Xs[lk + 2 * i + 1][lr] = xr[i][1];
            Ws[lk + 2 * i][lr] = wr[i][0]; Ws[lk + 2 * i + 1][lr] = wr[i][1];
        }
        __syncthreads();
        if (k0 + KT < K) {                                                // the next tile's operands travel while this one is multiplied
#pragma unroll
            for (int i = 0; i < RT; ++i) { xr[i] = xp[(k0 + KT) / 2 + i]; wr[i] = wp[(k0 + KT) / 2 + i]; }
        }
#pragma unroll 8
        for (int kk = 0; kk < KT; ++kk) {
            double xa[RT], wb[RT];
#pragma unroll
            for (int i = 0; i < RT; ++i) { xa[i] = Xs[kk][ty * RT + i]; wb[i] = Ws[kk][tx * RT + i]; }
#pragma unroll
            for (int i = 0; i < RT; ++i)
#pragma unroll
                for (int j = 0; j < RT; ++j) acc[i][j] = fma(xa[i], wb[j], acc[i][j]);
        }
    }
#pragma unroll
    for (int i = 0; i < RT; ++i) {
        const int r = r0 + ty * RT + i;
        if (r >= M) continue;
#pragma unroll
        for (int j = 0; j < RT; ++j) {
            const int nn = n0 + tx * RT + j;
            if (nn >= N) continue;
            double v = acc[i][j] + (bias ? bias[(size_t)l * N + nn] : 0.0);
            if (ACT == 2) v = v > 0.0 ? v : 0.2 * v;
            if (y64) y64[(size_t)r * ldy + (size_t)l * N + nn] = v;
            if (y32) y32[(size_t)r * ldy + (size_t)l * N + nn] = (float)v;
        }
    }
}

// A handful of rows (one streamed window, the CVAE branch's decoder calls: M <= 16): one wave per (row, two output columns), lanes stride K,
// wave reduction - a single round of loads instead of a chain of K tiles with two barriers each (M = 1: 2 x ~16 us on the tiled kernel)
template <int ACT>
__global__ __launch_bounds__(256) void mocha_linear_f64_rows(const double* __restrict__ X, int ldx, int xcol, const double* __restrict__ W,
                                                             const double* __restrict__ bias, double* __restrict__ y64, float* __restrict__ y32,
                                                             int ldy, int N, int K) {
    const int l = blockIdx.z, m = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = blockIdx.x * 8 + 2 * wave;
    if (n >= N) return;
    const int n1 = n + 1 < N ? n + 1 : n;
    const double* x = X + (size_t)m * ldx + (size_t)l * xcol;
    const double* w0 = W + ((size_t)l * N + n) * K;
    const double* w1 = W + ((size_t)l * N + n1) * K;
    double a0 = 0.0, a1 = 0.0;
    for (int k = lane; k < K; k += 64) { const double xv = x[k]; a0 = fma(xv, w0[k], a0); a1 = fma(xv, w1[k], a1); }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { a0 += __shfl_xor(a0, o); a1 += __shfl_xor(a1, o); }
    if (lane < 2 && n + lane < N) {
        double v = (lane ? a1 : a0) + (bias ? bias[(size_t)l * N + n + lane] : 0.0);
        if (ACT == 2) v = v > 0.0 ? v : 0.2 * v;
        if (y64) y64[(size_t)m * ldy + (size_t)l * N + n + lane] = v;
        if (y32) y32[(size_t)m * ldy + (size_t)l * N + n + lane] = (float)v;
    }
}

hipError_t launch_linear_f64(const double* X, int ldx, int xcol, const double* W, const double* bias, double* y64, float* y32, int ldy,
                             int M, int N, int K, int L, int act, hipStream_t s) {
    if (M <= 0) return hipSuccess;
    if (N < 1 || K < 32 || (K & 31) || L < 1 || (act != 0 && act != 2) || (!y64 && !y32) || (ldx & 1) || (xcol & 1)) return hipErrorInvalidValue;      // 16-byte row loads
    if (M <= 16) {
        const dim3 grid((N + 7) / 8, M, L);
        if (act == 2) hipLaunchKernelGGL(mocha_linear_f64_rows<2>, grid, dim3(256), 0, s, X, ldx, xcol, W, bias, y64, y32, ldy, N, K);
        else hipLaunchKernelGGL(mocha_linear_f64_rows<0>, grid, dim3(256), 0, s, X, ldx, xcol, W, bias, y64, y32, ldy, N, K);
        return hipGetLastError();
    }
    const long long big = (long long)((N + 63) / 64) * ((M + 63) / 64) * L;
    const bool small = big < 512;                                   // fewer than 512 tiles of 64 x 64: 32 x 32 tiles (4 x the workgroups)
    if (small) {
        const dim3 grid((N + 31) / 32, (M + 31) / 32, L);
        if (act == 2) hipLaunchKernelGGL((mocha_linear_f64<2, 2>), grid, dim3(256), 0, s, X, ldx, xcol, W, bias, y64, y32, ldy, M, N, K);
        else hipLaunchKernelGGL((mocha_linear_f64<0, 2>), grid, dim3(256), 0, s, X, ldx, xcol, W, bias, y64, y32, ldy, M, N, K);
    } else {
        const dim3 grid((N + 63) / 64, (M + 63) / 64, L);
        if (act == 2) hipLaunchKernelGGL((mocha_linear_f64<2, 4>), grid, dim3(256), 0, s, X, ldx, xcol, W, bias, y64, y32, ldy, M, N, K);
        else hipLaunchKernelGGL((mocha_linear_f64<0, 4>), grid, dim3(256), 0, s, X, ldx, xcol, W, bias, y64, y32, ldy, M, N, K);
    }
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------
// window_sums: operand of the joint temporal conv fused with AvgPool2d((4,1)) (blocks.py:112-118, model.py:47):
//   u[(b,t',p)][dt*C + c] = 1/4 * sum_{j<4} y[(b, refl(4t'+j+dt-2, 60), p)][c],   dt = 0..4
// on the C = 256 gcn outputs, or - with the gcn conv folded into the temporal conv's weights (mocha_api.cpp: emb.Wc) - on the
// C = 192 adjacency-mixed inputs of the gcn conv.
// The five overlapping 4-frame windows of one output row share 8 input frames, and consecutive output rows share 4.
// One thread owns a (window, body part, channel quad) and walks the 15 output frames with a sliding register window of 8
// input frames (4 new ones per step), so every input frame is fetched once (PMC had shown 2x the input bytes from HBM when
// each output row fetched its own 8 frames).
// ---------------------------------------------------------------------------------------
template <int CQ /* channel quads per row: 64 (256 channels) or 48 (192) */>
__global__ __launch_bounds__(256) void mocha_window_sums(const float* __restrict__ y, float* __restrict__ u, int rows /*B*15*6*/) {
    const int gid = blockIdx.x * 256 + threadIdx.x;
    const int bp = gid / CQ, q = gid - bp * CQ;         // bp = b * 6 + pp
    if (bp * 15 >= rows) return;
    const int b = bp / 6, pp = bp - b * 6;
    const f32x4* yb = reinterpret_cast<const f32x4*>(y) + ((size_t)b * 60 * 6 + pp) * CQ + q;
    f32x4* ub = reinterpret_cast<f32x4*>(u) + ((size_t)b * 15 * 6 + pp) * 5 * CQ + q;
    auto frame = [&](int t) __attribute__((always_inline)) {
        t = t < 0 ? -t : t;
        t = t > 59 ? 118 - t : t;                        // reflect padding (blocks.py:112-118)
        return yb[(size_t)t * 6 * CQ];
    };
    f32x4 f[8];
#pragma unroll
    for (int i = 0; i < 4; ++i) f[4 + i] = frame(-2 + i);
#pragma unroll
    for (int t15 = 0; t15 < 15; ++t15) {
#pragma unroll
        for (int i = 0; i < 4; ++i) { f[i] = f[4 + i]; f[4 + i] = frame(4 * t15 + 2 + i); }
        f32x4* ur = ub + (size_t)t15 * 6 * 5 * CQ;
#pragma unroll
        for (int dt = 0; dt < 5; ++dt) ur[dt * CQ] = (((f[dt] + f[dt + 1]) + f[dt + 2]) + f[dt + 3]) * 0.25f;
    }
}

hipError_t launch_window_sums(const float* y, float* u, int rows, int channels, hipStream_t s) {
    if (rows <= 0) return hipSuccess;
    if (rows % 90 || (channels != 256 && channels != 192)) return hipErrorInvalidValue;          // whole windows: 15 frames x 6 parts
    const long long threads = (long long)(rows / 15) * (channels / 4);
    const dim3 grid((unsigned)((threads + 255) / 256));
    if (channels == 256) hipLaunchKernelGGL(mocha_window_sums<64>, grid, dim3(256), 0, s, y, u, rows);
    else hipLaunchKernelGGL(mocha_window_sums<48>, grid, dim3(256), 0, s, y, u, rows);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------
// bank utilities
// ---------------------------------------------------------------------------------------
// out[row] = ||x[row] - sub||^2 (sub may be null)
__global__ __launch_bounds__(256) void mocha_rownorm2(const float* __restrict__ x, const float* __restrict__ sub, float* __restrict__ out,
                                                      int cols) {
    __shared__ float red[4];
    const size_t row = blockIdx.x;
    const f32x4* xr = reinterpret_cast<const f32x4*>(x + row * cols);
    float a = 0.f;
    for (int i = threadIdx.x; i < cols / 4; i += 256) {
        f32x4 v = xr[i];
        if (sub) v -= reinterpret_cast<const f32x4*>(sub)[i];
        a = fmaf(v[0], v[0], a); a = fmaf(v[1], v[1], a); a = fmaf(v[2], v[2], a); a = fmaf(v[3], v[3], a);
    }
    a = wave_sum(a);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = a;
    __syncthreads();
    if (threadIdx.x == 0) out[row] = (red[0] + red[1]) + (red[2] + red[3]);
}

hipError_t launch_rownorm2(const float* x, const float* sub, float* out, int64_t rows, int cols, hipStream_t s) {
    if (rows <= 0) return hipSuccess;
    if (cols % 4) return hipErrorInvalidValue;
    hipLaunchKernelGGL(mocha_rownorm2, dim3((unsigned)rows), dim3(256), 0, s, x, sub, out, cols);
    return hipGetLastError();
}

// out[row] = x[row] - sub   (queries centred on the bank centroid); flat grid, one thread per float4
__global__ __launch_bounds__(256) void mocha_sub_rows(const float* __restrict__ x, const float* __restrict__ sub, float* __restrict__ out,
                                                      int cols4, long long total4) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total4) return;
    reinterpret_cast<f32x4*>(out)[i] = reinterpret_cast<const f32x4*>(x)[i] - reinterpret_cast<const f32x4*>(sub)[i % cols4];
}

hipError_t launch_sub_rows(const float* x, const float* sub, float* out, int64_t rows, int cols, hipStream_t s) {
    if (rows <= 0) return hipSuccess;
    if (cols % 4) return hipErrorInvalidValue;
    const long long total4 = (long long)rows * (cols / 4);
    hipLaunchKernelGGL(mocha_sub_rows, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, s, x, sub, out, cols / 4, total4);
    return hipGetLastError();
}

// One workgroup per row: x - centre as two stacked bf16 planes (or as fp32) and the row's statistics for the many-query selection
// (match_select2.hip): ||x - c||^2 and the squared norm of what the planes leave out.  Thread t takes the 4-element pieces t, t + 512, ...;
// wave sums, then the eight waves in order: the statistics are reproducible.
__global__ __launch_bounds__(512) void mocha_center_rows(const float* __restrict__ x, const float* __restrict__ c, unsigned short* __restrict__ planes,
                                                         float* __restrict__ out32, float* __restrict__ qstat, long long plane_stride, int cols4) {
    __shared__ float red[2][8];
    const size_t row = blockIdx.x;
    const f32x4* xr = reinterpret_cast<const f32x4*>(x) + row * cols4;
    const f32x4* cr = reinterpret_cast<const f32x4*>(c);
    float qn = 0.f, dn = 0.f;
    for (int i = threadIdx.x; i < cols4; i += 512) {
        const f32x4 z = xr[i] - cr[i];
        qn = fmaf(z[0], z[0], qn); qn = fmaf(z[1], z[1], qn); qn = fmaf(z[2], z[2], qn); qn = fmaf(z[3], z[3], qn);
        if (out32) reinterpret_cast<f32x4*>(out32)[row * cols4 + i] = z;
        else {
            const u32x2_t w0 = {bf16_bits(z[0]) | (bf16_bits(z[1]) << 16), bf16_bits(z[2]) | (bf16_bits(z[3]) << 16)};
            f32x4 r = z - f32x4{__uint_as_float(w0[0] << 16), __uint_as_float(w0[0] & 0xffff0000u), __uint_as_float(w0[1] << 16), __uint_as_float(w0[1] & 0xffff0000u)};
            reinterpret_cast<u32x2_t*>(planes)[row * cols4 + i] = w0;
            if (plane_stride > 0) {
                const u32x2_t w1 = {bf16_bits(r[0]) | (bf16_bits(r[1]) << 16), bf16_bits(r[2]) | (bf16_bits(r[3]) << 16)};
                r -= f32x4{__uint_as_float(w1[0] << 16), __uint_as_float(w1[0] & 0xffff0000u), __uint_as_float(w1[1] << 16), __uint_as_float(w1[1] & 0xffff0000u)};
                reinterpret_cast<u32x2_t*>(planes + plane_stride)[row * cols4 + i] = w1;
            }
            dn = fmaf(r[0], r[0], dn); dn = fmaf(r[1], r[1], dn); dn = fmaf(r[2], r[2], dn); dn = fmaf(r[3], r[3], dn);
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { qn += __shfl_xor(qn, o); dn += __shfl_xor(dn, o); }
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = qn; red[1][threadIdx.x >> 6] = dn; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float a = red[0][0], d = red[1][0];
        for (int w = 1; w < 8; ++w) { a += red[0][w]; d += red[1][w]; }
        float* qo = qstat + (size_t)row * 2 * QSTAT_PARTS;      // the whole row in part 0
        qo[0] = a; qo[1] = d;
        for (int k = 1; k < QSTAT_PARTS; ++k) { qo[2 * k] = 0.f; qo[2 * k + 1] = 0.f; }
    }
}

hipError_t launch_center_rows(const float* x, const float* centre, void* planes, int nplanes, float* out32, float* qstat, int64_t rows, int cols, hipStream_t s) {
    if (rows <= 0) return hipSuccess;
    if (cols % 4 || !qstat || (planes == nullptr) == (out32 == nullptr) || (planes && nplanes != 1 && nplanes != 2)) return hipErrorInvalidValue;
    hipLaunchKernelGGL(mocha_center_rows, dim3((unsigned)rows), dim3(512), 0, s, x, centre, (unsigned short*)planes, out32, qstat,
                       nplanes == 2 ? (long long)rows * cols : 0ll, cols / 4);
    return hipGetLastError();
}

__global__ __launch_bounds__(256) void mocha_gather_rows(const float* __restrict__ src, const int32_t* __restrict__ idx,
                                                         float* __restrict__ out, int cols4, long long nrows) {
    const size_t q = blockIdx.x;
    long long r = idx[q];                              // caller-supplied indices are clamped into the bank: never out of bounds
    r = r < 0 ? 0 : (r >= nrows ? nrows - 1 : r);
    const f32x4* s = reinterpret_cast<const f32x4*>(src) + (size_t)r * cols4;
    f32x4* o = reinterpret_cast<f32x4*>(out) + q * cols4;
    for (int i = threadIdx.x; i < cols4; i += 256) o[i] = s[i];
}

// Soft context matching over the k neighbours of a top-k query: out[q] = sum_j softmax_j(-dist[q][j] / temperature) src[idx[q][j]]
// (neighbours with idx < 0 - a bank with fewer than k rows - are left out).  k <= 64.
__global__ __launch_bounds__(256) void mocha_gather_blend(const float* __restrict__ src, const int32_t* __restrict__ idx,
                                                          const float* __restrict__ dist, float inv_temp, float* __restrict__ out, int k,
                                                          int cols4, long long nrows) {
    __shared__ float w[64];
    __shared__ long long rows[64];
    const size_t q = blockIdx.x;
    if (threadIdx.x == 0) {
        float mx = -INFINITY;
        for (int j = 0; j < k; ++j) {
            const long long r = idx[q * k + j];
            rows[j] = (r >= 0 && r < nrows) ? r : -1;
            if (rows[j] >= 0) mx = fmaxf(mx, -dist[q * k + j] * inv_temp);
        }
        float sum = 0.f;
        for (int j = 0; j < k; ++j) { w[j] = rows[j] >= 0 ? __expf(-dist[q * k + j] * inv_temp - mx) : 0.f; sum += w[j]; }
        for (int j = 0; j < k; ++j) w[j] = sum > 0.f ? w[j] / sum : 0.f;
    }
    __syncthreads();
    f32x4* o = reinterpret_cast<f32x4*>(out) + q * cols4;
    for (int i = threadIdx.x; i < cols4; i += 256) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int j = 0; j < k; ++j)
            if (rows[j] >= 0) acc += w[j] * (reinterpret_cast<const f32x4*>(src) + (size_t)rows[j] * cols4)[i];
        o[i] = acc;
    }
}

hipError_t launch_gather_blend(const float* src, const int32_t* idx, const float* dist, float temperature, float* out, int Q, int k,
                               int cols, int64_t nrows, hipStream_t s) {
    if (Q <= 0) return hipSuccess;
    if (cols % 4 || nrows < 1 || k < 1 || k > 64 || !(temperature > 0.f)) return hipErrorInvalidValue;
    hipLaunchKernelGGL(mocha_gather_blend, dim3(Q), dim3(256), 0, s, src, idx, dist, 1.0f / temperature, out, k, cols / 4, (long long)nrows);
    return hipGetLastError();
}

hipError_t launch_gather_rows(const float* src, const int32_t* idx, float* out, int Q, int cols, int64_t nrows, hipStream_t s) {
    if (Q <= 0) return hipSuccess;
    if (cols % 4 || nrows < 1) return hipErrorInvalidValue;
    hipLaunchKernelGGL(mocha_gather_rows, dim3(Q), dim3(256), 0, s, src, idx, out, cols / 4, (long long)nrows);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------
// column statistics over the bank entries (bank build, compute_cnt_norm.py:174-175):
//   mean[j] = mean_n x[n][j] ;  std[j] = sqrt(mean_n (x[n][j] - mean[j])^2)      (numpy default: population std)
// One thread per column, rows strided over the 4 waves of a workgroup is not needed: a column block of 64 adjacent
// columns is read as coalesced 256-byte row segments; fp64 accumulation makes the result independent of N's size.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void mocha_column_stats(const float* __restrict__ x, long long N, int cols,
                                                          float* __restrict__ mean, float* __restrict__ sd) {
    __shared__ double s1[4][64], s2[4][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int col = blockIdx.x * 64 + lane;
    if (col >= cols) return;
    double a = 0.0;
    for (long long n = w; n < N; n += 4) a += (double)x[(size_t)n * cols + col];
    s1[w][lane] = a;
    __syncthreads();
    const double m = ((s1[0][lane] + s1[1][lane]) + (s1[2][lane] + s1[3][lane])) / (double)N;
    if (!sd) {                                         // mean only (bank centroid); uniform per launch
        if (w == 0) mean[col] = (float)m;
        return;
    }
    double q = 0.0;
    for (long long n = w; n < N; n += 4) { const double d = (double)x[(size_t)n * cols + col] - m; q += d * d; }
    s2[w][lane] = q;
    __syncthreads();
    if (w == 0) {
        mean[col] = (float)m;
        sd[col] = (float)sqrt(((s2[0][lane] + s2[1][lane]) + (s2[2][lane] + s2[3][lane])) / (double)N);
    }
}

// column mean in two deterministic stages (the bank centroid is on the per-step path of the demo pair): CM_CHUNKS row
// chunks per 64-column block accumulate in fp64, a second kernel adds the partials in a fixed order
static constexpr int CM_CHUNKS = 16;

__global__ __launch_bounds__(256) void mocha_column_mean_part(const float* __restrict__ x, long long N, int cols,
                                                              double* __restrict__ part /*[CM_CHUNKS][4][cols]*/) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int col = blockIdx.x * 64 + lane, chunk = blockIdx.y;
    const long long per = (N + CM_CHUNKS - 1) / CM_CHUNKS;
    const long long lo = chunk * per, hi = (lo + per) < N ? (lo + per) : N;
    double a = 0.0;
    for (long long n = lo + w; n < hi; n += 4) a += (double)x[(size_t)n * cols + col];
    part[((size_t)chunk * 4 + w) * cols + col] = a;
}

__global__ __launch_bounds__(256) void mocha_column_mean_fin(const double* __restrict__ part, long long N, int cols,
                                                             float* __restrict__ mean) {
    const int col = blockIdx.x * 256 + threadIdx.x;
    if (col >= cols) return;
    double a = 0.0;
    for (int k = 0; k < CM_CHUNKS * 4; ++k) a += part[(size_t)k * cols + col];
    mean[col] = (float)(a / (double)N);
}

size_t column_mean_scratch_doubles(int cols) { return (size_t)CM_CHUNKS * 4 * cols; }

hipError_t launch_column_mean(const float* x, int64_t N, int cols, float* mean, double* scratch, hipStream_t s) {
    if (N <= 0 || cols <= 0 || cols % 64) return hipErrorInvalidValue;
    hipLaunchKernelGGL(mocha_column_mean_part, dim3(cols / 64, CM_CHUNKS), dim3(256), 0, s, x, (long long)N, cols, scratch);
    hipLaunchKernelGGL(mocha_column_mean_fin, dim3((cols + 255) / 256), dim3(256), 0, s, scratch, (long long)N, cols, mean);
    return hipGetLastError();
}

hipError_t launch_column_stats(const float* x, int64_t N, int cols, float* mean, float* sd, hipStream_t s) {
    if (N <= 0 || cols <= 0) return hipErrorInvalidValue;
    if (cols % 64) return hipErrorInvalidValue;
    hipLaunchKernelGGL(mocha_column_stats, dim3(cols / 64), dim3(256), 0, s, x, (long long)N, cols, mean, sd);
    return hipGetLastError();
}

}  // namespace mocha
