// fp32 GEMM on the fp16 matrix pipe of gfx950 with TWO planes per operand and THREE passes per product (option "gemm_f16x2", default off):
//
//   x = (h0 + h1) / S,  h0 = fp16(S x), h1 = fp16(S x - h0)            (round to nearest even: 22 significant bits; S a power of two)
//   a b ~ a0 b0 + (a0 b1 + a1 b0)                                      (the dropped a1 b1 is 2^-22 |ab|, like the planes' own truncation)
//
// Every fp16 x fp16 product is exact in the fp32 accumulator of v_mfma_f32_32x32x16_f16.  Measured against float64 the result is, at the
// path's K (256 ... 1 280), MORE accurate than both fp32 engines of the library (gemm_f32.hip's exact pipe and gemm_x3.hip's three bf16
// planes / six passes): what a K-long fp32 dot product loses is dominated by the accumulator's own roundings, and three passes have half
// as many as six; at K <= 64 the planes' 2^-22 shows, still below the exact pipe (tools/f16x2_sim.py; tests/test_gemm_f16x2.py;
// profiles/r05/r_f16x2_prototype_gemm_bench.txt) - at 0.62-0.75 of the six-pass time.
// It is NOT a per-product-faithful fp32 multiply (2^-22 instead of 2^-24 per product), which is why it is an option.
//
// Scales (exact powers of two, so the result does not depend on them while nothing leaves fp16's normal range):
//   * weights: per output row, from the row's largest magnitude (mocha_h2_wscale at pack time); the epilogue multiplies column n by 1 / S_w[n];
//   * activations: one scale per WINDOW (the path's independent unit: GemmParams::rows_per_win consecutive rows of the launch), from a bound on
//     the window's |A| the caller passes as a vector in device memory (GemmParams::a_amax): the window's largest scaled element lands in
//     [2^14, 2^15).  The producing GEMM's epilogue maintains such a vector (GemmParams::c_amax: per window the largest magnitude it stores),
//     bounds that follow from the arithmetic serve the other producers - every operator between two GEMMs of the path stays inside a window
//     (softmax-weighted rows are bounded by the window's value rows, an instance-normalised token by (n - 1) / sqrt(n), a column-normalised
//     adjacency mix by its input) - mocha_absmax the rest.  A window's result therefore does not depend on the other windows of a batch.
//     fp16 keeps 22 bits over 2^18 of range: rows more than five decades below THEIR WINDOW's largest magnitude lose digits (absolute
//     error 2^-40 of that magnitude) - the fp32 engines keep them.
//
// Kernel: the one-shot grid of gemm_x3.hip's plane GEMM (that file has the reasoning for the tiling, the LDS image, the loaders and the counted waits)
// with two planes per operand: 128 x 128 / 64 x 128 / 128 x 64 / 64 x 64 tiles, K step 16, per step and wave 8 ds_read_b128 feed 12
// MFMAs (TM = TN = 2), 12 TM split instructions (v_cvt_pk_f16_f32, v_cvt_f32_f16, subtract) interleaved two (TN = 1: four) per MFMA; 34 KB of LDS.
#include "kernels.h"
#include "device_utils.h"
#include <type_traits>
#include <algorithm>

namespace mocha {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
#ifndef H2_WPE
#define H2_WPE 3          // workgroups (of four waves) per CU the register budget is set for; 4 was tried: see tools/experiments/README.md
#endif
static constexpr int NPL = 2;                            // planes per operand
__device__ __forceinline__ unsigned cvt_pk_f16(float a, float b) { const f32x2 v = {a, b}; return __builtin_bit_cast(unsigned, __builtin_convertvector(v, h16x2)); }
__device__ __forceinline__ float f16_lo(unsigned p) { return (float)__builtin_bit_cast(h16x2, p)[0]; }
__device__ __forceinline__ float f16_hi(unsigned p) { return (float)__builtin_bit_cast(h16x2, p)[1]; }
// four floats (already scaled) -> two planes of four fp16 (8 bytes each)
__device__ __forceinline__ void f16_split4(const f32x4 v, u32x2 (&out)[2]) {
    const unsigned p01 = cvt_pk_f16(v[0], v[1]), p23 = cvt_pk_f16(v[2], v[3]);
    out[0][0] = p01; out[0][1] = p23;
    out[1][0] = cvt_pk_f16(v[0] - f16_lo(p01), v[1] - f16_hi(p01)); out[1][1] = cvt_pk_f16(v[2] - f16_lo(p23), v[3] - f16_hi(p23));
}
static constexpr int F16_PA[3] = {1, 0, 0}, F16_PB[3] = {0, 1, 0};      // a1 b0, a0 b1, a0 b0 (low-order products first)

static constexpr int XN = 128, XK = 16;                  // tile width, K step; tile height = 64 TM rows
static constexpr int XA_HALF = 128 * 8 + 32;            // fp16 per k half of an A plane (2 KB + 64 B)
static constexpr int XA_PLANE = 2 * XA_HALF;            // 2176 fp16
static constexpr int XB_PLANE = 128 * 16;               // 2048 fp16, [k half][row][8]
static constexpr int XB_OFF = NPL * XA_PLANE;             // B planes follow the A planes of a stage
static constexpr int X_STAGE = XB_OFF + NPL * XA_PLANE;   // 8 704 fp16 = 17 408 B (the B planes use the padded A layout in LDS)
static constexpr int XW_BLOCK = NPL * XB_PLANE;           // packed weights per (n tile, k step): 4 096 fp16 = 8 KB
// tile width 64 TN: the B planes of a stage hold 64 TN rows per k half (TN = 2: the layout above; TN = 1: one 64-row half of a packed block)
template <int TN> struct XT {
    static constexpr int TILE_N = 64 * TN;
    static constexpr int B_HALF = TILE_N * 8 + 32;
    static constexpr int B_PLANE = 2 * B_HALF;
    static constexpr int STAGE = XB_OFF + NPL * B_PLANE;
};
static_assert(XT<2>::STAGE == X_STAGE && XT<2>::B_HALF == XA_HALF, "TN = 2 is the 128-wide layout");

__device__ __forceinline__ float h2_lrelu(float x) { return x > 0.f ? x : 0.2f * x; }
__device__ __forceinline__ float h2_gelu(float x) { return mocha_gelu(x); }

// power-of-two scale that puts a largest magnitude `amax` into [2^14, 2^15): S = 2^(14 - floor(log2 amax)); 1 for zero, denormal-range, infinite
// or NaN bounds (whatever is multiplied then is zero or already lost).  Returns S, *inv = 1 / S (both exact).
__host__ __device__ __forceinline__ float h2_scale(float amax, float* inv) {
    unsigned u;
    __builtin_memcpy(&u, &amax, 4);
    const int e = (int)((u >> 23) & 0xffu);
    if (e < 16 || e > 250) { *inv = 1.f; return 1.f; }
    const unsigned us = (unsigned)(268 - e) << 23, ui = (unsigned)(e - 14) << 23;
    float sc; __builtin_memcpy(&sc, &us, 4); __builtin_memcpy(inv, &ui, 4);
    return sc;
}

// per weight row: 1 / S_w[n] (one wave per row)
__global__ __launch_bounds__(256) void mocha_h2_wscale(const float* __restrict__ W, int N, int K, float* __restrict__ w_inv) {
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (n >= N) return;
    float m = 0.f;
    for (int k = lane; k < K; k += 64) m = fmaxf(m, fabsf(W[(size_t)n * K + k]));
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if (lane == 0) { float inv; (void)h2_scale(m, &inv); w_inv[n] = inv; }
}

// out[w] = max(out[w], mul * (largest magnitude of window w's `per` floats) + add) for nwin consecutive windows of x (out zeroed by the caller;
// mul > 0, add >= 0: the bound of a linear map of x with row L1 norm <= mul and bias magnitudes <= add).  grid = (parts, nwin): a window's floats
// are cut into `parts` slices; atomic max on the bits of a non-negative float (a handful per address)
__global__ __launch_bounds__(256) void mocha_absmax(const float* __restrict__ x, long long per, float* __restrict__ out, float mul, float add) {
    const long long lo = per * blockIdx.x / gridDim.x, hi = per * (blockIdx.x + 1) / gridDim.x;
    const float* xw = x + (long long)blockIdx.y * per + lo;
    const long long n = hi - lo;
    float m = 0.f;
    // 16-byte loads from the first aligned element on; the up to three elements in front of it and behind the last quad one by one
    const long long head = std::min<long long>(n, (long long)((16 - (reinterpret_cast<uintptr_t>(xw) & 15)) & 15) >> 2);
    const float* xa = xw + head;
    const long long n4 = (n - head) >> 2;
    for (long long i = threadIdx.x; i < n4; i += 256) {
        const f32x4 v = reinterpret_cast<const f32x4*>(xa)[i];
        m = fmaxf(fmaxf(m, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
    }
    if (threadIdx.x < head) m = fmaxf(m, fabsf(xw[threadIdx.x]));
    if (threadIdx.x < ((n - head) & 3)) m = fmaxf(m, fabsf(xa[n4 * 4 + threadIdx.x]));
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    __shared__ float wm[4];
    if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) atomicMax(reinterpret_cast<unsigned*>(out + blockIdx.y), __float_as_uint(fmaf(mul, fmaxf(fmaxf(wm[0], wm[1]), fmaxf(wm[2], wm[3])), add)));
}
hipError_t launch_absmax(const float* x, long long nwin, long long per, float* out, hipStream_t s, float mul, float add) {
    if (nwin <= 0 || per <= 0) return hipSuccess;
    if ((reinterpret_cast<uintptr_t>(x) & 3) != 0 || nwin > 65535 || !(mul > 0.f) || !(add >= 0.f)) return hipErrorInvalidValue;
    // enough workgroups to fill the chip: one per window when there are many, slices of at least 4 096 floats otherwise
    long long parts = std::max<long long>(1, std::min<long long>(2048 / nwin, per / 4096));
    hipLaunchKernelGGL(mocha_absmax, dim3((unsigned)parts, (unsigned)nwin), dim3(256), 0, s, x, per, out, mul, add);
    return hipGetLastError();
}
// out[i] = table[clamp(idx[i])] (the decoder's per-window value bound = its matched entry's)
__global__ __launch_bounds__(256) void mocha_gather_f32(const float* __restrict__ table, const int32_t* __restrict__ idx, long long rows, float* __restrict__ out, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    long long r = idx[i];
    r = r < 0 ? 0 : r >= rows ? rows - 1 : r;
    out[i] = table[r];
}
hipError_t launch_gather_f32(const float* table, const int32_t* idx, long long rows, float* out, int n, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(mocha_gather_f32, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, table, idx, rows, out, n);
    return hipGetLastError();
}

// W [N][K] fp32 -> packed planes [n tile][k step][plane][k half][128 rows][8 fp16] of S_w[n] W (w_inv from mocha_h2_wscale).  One workgroup
// per (n tile, k step) block: thread = (row, k half) reads 32 bytes and writes one 16-byte piece per plane.
__global__ __launch_bounds__(256) void mocha_pack_h2(const float* __restrict__ W, const float* __restrict__ w_inv, int N, int K, unsigned short* __restrict__ out) {
    const int ksteps = K / XK;
    const int nt = blockIdx.x / ksteps, ks = blockIdx.x - nt * ksteps;
    const int r = threadIdx.x >> 1, h = threadIdx.x & 1;
    const int n = nt * XN + r;
    const int k = ks * XK + 8 * h;
    f32x4 lo = {0.f, 0.f, 0.f, 0.f}, hi = lo;
    if (n < N) {
        const float sw = 1.f / w_inv[n];                 // exact: a power of two
        lo = *reinterpret_cast<const f32x4*>(W + (size_t)n * K + k) * sw;
        hi = *reinterpret_cast<const f32x4*>(W + (size_t)n * K + k + 4) * sw;
    }
    u32x2 a[NPL], b[NPL];
    f16_split4(lo, a); f16_split4(hi, b);
    unsigned short* blk = out + (size_t)blockIdx.x * XW_BLOCK;
#pragma unroll
    for (int q = 0; q < NPL; ++q) {
        const u32x4 v = {a[q][0], a[q][1], b[q][0], b[q][1]};
        *reinterpret_cast<u32x4*>(blk + q * XB_PLANE + h * 1024 + r * 8) = v;
    }
}

size_t gemm_h2_packed_elems(int N, int K) { return (size_t)((N + XN - 1) / XN) * (K / XK) * XW_BLOCK; }

hipError_t launch_pack_h2(const float* W, int N, int K, unsigned short* out, float* w_inv, hipStream_t s) {
    if (K % XK != 0 || N <= 0) return hipErrorInvalidValue;
    const long long blocks = (long long)((N + XN - 1) / XN) * (K / XK);
    if (blocks > 0x7fffffffll) return hipErrorInvalidValue;
    hipLaunchKernelGGL(mocha_h2_wscale, dim3((unsigned)((N + 3) / 4)), dim3(256), 0, s, W, N, K, w_inv);
    hipLaunchKernelGGL(mocha_pack_h2, dim3((unsigned)blocks), dim3(256), 0, s, W, w_inv, N, K, out);
    return hipGetLastError();
}

// per-window maxima of what a tile stores: a thread keeps one register per window the tile touches (at most three), the waves reduce them ONCE per
// tile (an LDS atomic per stored value - 32 lanes on one address, 16 times per thread - cost the 9 876-tile qkv launch 64 us), the workgroup
// leaves them in the output's bound vector with at most three global atomics
__device__ __forceinline__ void h2_track(float (&mw)[3], int wl, float m) {
    mw[0] = wl == 0 ? fmaxf(mw[0], m) : mw[0];
    mw[1] = wl == 1 ? fmaxf(mw[1], m) : mw[1];
    mw[2] = wl == 2 ? fmaxf(mw[2], m) : mw[2];
}
__device__ __forceinline__ void h2_amax_out(float* c_amax, float (&mw)[3], unsigned* wmax /*LDS, zeroed at kernel start*/, int w0, int nwin, int lane, int tid) {
    if (!c_amax) return;                             // uniform
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        float m = mw[k];
        for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
        if (lane == 0 && m > 0.f) atomicMax(&wmax[k], __float_as_uint(m));
    }
    __syncthreads();
    if (tid < 3 && w0 + tid < nwin && wmax[tid]) atomicMax(reinterpret_cast<unsigned*>(c_amax) + w0 + tid, wmax[tid]);
}

// LRELU: LeakyReLU(0.2) on the activations as they are split; GATHER: temporal-conv gather (kernels.h) instead of plain rows.
// Compile-time so that a K step is one basic block the scheduler can interleave.
// TM: 32-row MFMA blocks per wave: 2 = the 128-row tile; 1 = a 64-row tile (four waves of 32 x 64) for mid-size launches
// (a few dozen to a few hundred windows), where 128-row tiles would leave most workgroup slots empty.
// TN: 32-column MFMA blocks per wave: 2 = the 128-wide tile; 1 = a 128 x 64 tile (waves of 64 x 32) for N = 64 / 192 (to_mot's joint
// block), where a padded 128-wide tile would idle half the pipe.  (A 128 x 256 tile, TN = 4, was measured and not kept:
// tools/experiments/gemm_x3_tile_128x256.patch.txt.)
template <bool LRELU, bool GATHER, int TM, int TN>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(H2_WPE, H2_WPE))) void mocha_gemm_h2(GemmParams p) {
    constexpr int TILE_M = TM * 64;
    constexpr int TILE_N = XT<TN>::TILE_N, B_HALF = XT<TN>::B_HALF, B_PLANE = XT<TN>::B_PLANE, STAGE = XT<TN>::STAGE;
    extern __shared__ __attribute__((aligned(16))) unsigned short h2_sm[];          // [2][STAGE]
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hh = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    __shared__ float h2_rowinv[128];                // 1 / S_a of the tile's rows (their windows' scales)
    __shared__ unsigned h2_wmax[4];                 // bits of the largest magnitude stored per window the tile touches (at most three: rows_per_win >= 64)

    const int n_tiles = (p.N + TILE_N - 1) / TILE_N;
    const int m_tiles = (p.M + TILE_M - 1) / TILE_M;
    const int bid = blockIdx.x;
    int mt, nt;
    if (m_tiles >= 8) {                 // XCD-aware order: the n-tiles of one m-tile share an XCD (bid % 8)
        const int grp = bid / (8 * n_tiles);
        const int rem = bid - grp * 8 * n_tiles;
        mt = grp * 8 + (rem & 7);
        nt = rem >> 3;
    } else {
        mt = bid / n_tiles;
        nt = bid - mt * n_tiles;
    }
    if (mt >= m_tiles) return;
    const int m0 = mt * TILE_M, n0 = nt * TILE_N;
    // K split over gridDim.z (the matcher's 23 040-long contraction): this workgroup takes steps s0 .. s0 + nsteps - 1 and writes raw
    // partial sums to slab blockIdx.z; the host guarantees at least two steps per slab
    const int steps_total = p.K / XK;
    const int per = (steps_total + p.ksplit - 1) / p.ksplit;
    const int s0 = blockIdx.z * per;
    const int nsteps = (s0 + per) <= steps_total ? per : steps_total - s0;

    // ---- A loader: four lanes cover the 64-byte row segment of a step; a thread takes rows lrow and lrow + 64
    const int lrow = tid >> 2;
    const int lc = tid & 3;
    const __amdgpu_buffer_rsrc_t rsA = make_rsrc(GATHER ? p.A : p.A + (size_t)(m0 < p.M ? m0 : 0) * p.lda);
    int a_rb[TM], a_t[TM];
    unsigned a_off[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        int m = m0 + lrow + 64 * i;
        m = m < p.M ? m : p.M - 1;
        if (GATHER) {
            const int v = m % p.V;
            const int bt = m / p.V;
            a_t[i] = bt % p.T_out;
            a_rb[i] = (bt / p.T_out) * p.T_src * p.V + v;
            a_off[i] = 0;
        } else {
            a_rb[i] = m; a_t[i] = 0;
            a_off[i] = ((unsigned)(m - m0) * (unsigned)p.lda + lc * 4) * 4u;
        }
    }
    // windows: row m of the launch belongs to window m / rows_per_win; the tile's rows lie in windows w0 .. w0 + 2 at most, the first row of
    // window w0 + 1 / w0 + 2 is tile row wb1 / wb2
    const int w0 = m0 / p.rows_per_win;
    const int wb1 = (w0 + 1) * p.rows_per_win - m0, wb2 = wb1 + p.rows_per_win;
    const int nwin = (p.M + p.rows_per_win - 1) / p.rows_per_win;
    float sa[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int r = lrow + 64 * i;
        int w = w0 + (r >= wb1) + (r >= wb2);
        w = w < nwin ? w : nwin - 1;
        float inv;
        sa[i] = h2_scale(p.a_amax[w], &inv);
        if (lc == 0) h2_rowinv[r] = inv;
    }
    if (tid < 4) h2_wmax[tid] = 0u;
    f32x4 rset[2][TM];                              // step t's activations wait in set t & 1, fetched two steps ahead
    auto load_a = [&](int s, f32x4 (&ra)[TM]) __attribute__((always_inline)) {
        const int k0 = (s0 + s) * XK;
        if (!GATHER) {
#pragma unroll
            for (int i = 0; i < TM; ++i) ra[i] = bload(rsA, a_off[i], (unsigned)k0 * 4u);
        } else {
            // the row of tap k0 / Cc, recomputed every step (a handful of VALU instructions hidden between the MFMAs; no branch)
            const int tap = k0 / p.Cc;
            const int cin = k0 - tap * p.Cc;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                int tf = a_t[i] * p.stride + tap * p.tstep - p.pad;
                tf = tf < 0 ? -tf : tf;
                tf = tf >= p.T_full ? 2 * (p.T_full - 1) - tf : tf;
                const unsigned off = ((unsigned)(a_rb[i] + (tf >> p.tshift) * p.V) * (unsigned)p.lda + lc * 4) * 4u;
                ra[i] = bload(rsA, off, (unsigned)cin * 4u);
            }
        }
    };
    // plane q of (row, piece lc): k half lc >> 1, 8 bytes at (lc & 1)
    const int a_wr = (lc >> 1) * XA_HALF + lrow * 8 + (lc & 1) * 4;
    auto split_store = [&](const f32x4 (&ra)[TM], unsigned short* st) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            f32x4 v = ra[i];
            if (LRELU) { v[0] = h2_lrelu(v[0]); v[1] = h2_lrelu(v[1]); v[2] = h2_lrelu(v[2]); v[3] = h2_lrelu(v[3]); }
            u32x2 pl[NPL];
            { f32x4 w = v * sa[i]; w[0] = __builtin_amdgcn_fmed3f(w[0], -65504.f, 65504.f); w[1] = __builtin_amdgcn_fmed3f(w[1], -65504.f, 65504.f); w[2] = __builtin_amdgcn_fmed3f(w[2], -65504.f, 65504.f); w[3] = __builtin_amdgcn_fmed3f(w[3], -65504.f, 65504.f); f16_split4(w, pl); }
#pragma unroll
            for (int q = 0; q < NPL; ++q) *reinterpret_cast<u32x2*>(st + q * XA_PLANE + a_wr + i * 64 * 8) = pl[q];
        }
    };
    // ---- W: linear copy of the packed 8 KB block of (nt, step) into the stage
    const int wblock = TN == 1 ? nt >> 1 : nt;           // the packed image is in 128-column blocks; a 64-wide tile takes one half of one
    const __amdgpu_buffer_rsrc_t rsW = make_rsrc(p.Wh2 + ((size_t)wblock * steps_total + s0) * XW_BLOCK);
    auto dma_w = [&](int s, unsigned short* st) __attribute__((always_inline)) {
        if (TN == 2) {
#pragma unroll
            for (int j = 0; j < NPL; ++j)
                // piece j * 4 + wave of the packed block = (plane, k half, 64-row half): same padded halves as the A planes
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (__attribute__((address_space(3))) void*)(st + XB_OFF + ((j * 4 + wave) >> 2) * B_PLANE +
                                                         (((j * 4 + wave) >> 1) & 1) * B_HALF + ((j * 4 + wave) & 1) * 512), 16,
                                                         (unsigned)(j * 256 + tid) * 16u, (unsigned)s * (XW_BLOCK * 2u), 0, 0);
        } else {
            // four 1 KB pieces (plane, k half) of this tile's 64-row half: one per wave
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int pc = j * 4 + wave;                 // (plane, k half) = (pc >> 1, pc & 1)
                if (pc < 2 * NPL)                            // wave-uniform
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (__attribute__((address_space(3))) void*)(st + XB_OFF + (pc >> 1) * B_PLANE + (pc & 1) * B_HALF), 16,
                                                             (unsigned)(((pc >> 1) * 4 + (pc & 1) * 2 + (nt & 1)) * 64 + lane) * 16u, (unsigned)s * (XW_BLOCK * 2u), 0, 0);
            }
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // fragment of lane (row l31, k half hh): 16 bytes
    const int fa = hh * XA_HALF + (wm * TM * 32 + l31) * 8;
    const int fb = XB_OFF + hh * B_HALF + (wn * 32 * TN + l31) * 8;

    // prologue: step 0 into stage 0, step 1's activations into registers.  The counted waits below (and in the steps) rely on the
    // issue order of the copies relative to the register fetches; both are independent loads to the scheduler, so they are fenced.
    load_a(0, rset[0]);
    __builtin_amdgcn_sched_barrier(0);
    dma_w(0, h2_sm);
    __builtin_amdgcn_sched_barrier(0);
    split_store(rset[0], h2_sm);
    load_a(1, rset[1]);                             // K >= 32 (gemm_h2_supports)
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(TM) : "memory");     // step 1's TM fetches stay in flight

    // One K step.  FETCH_W: step s + 1 exists (its weights are copied and its activations split into the other stage);
    // FETCH_A: step s + 2 exists (its activations are fetched).  The three variants are straight-line code, so the compiler's own
    // vmcnt bookkeeping for the activation registers is exact: the split waits for the two oldest fetches only, not for the copy.
    auto step = [&](int s, auto parity, auto fetch_w, auto fetch_a) __attribute__((always_inline)) {
        constexpr bool FETCH_W = decltype(fetch_w)::value, FETCH_A = decltype(fetch_a)::value;
        constexpr int P = decltype(parity)::value;      // s & 1
        unsigned short* cur = h2_sm + P * STAGE;
        unsigned short* nxt = h2_sm + (P ^ 1) * STAGE;
        if (FETCH_W) dma_w(s + 1, nxt);             // first thing after the barrier: a whole step to land
        __builtin_amdgcn_sched_barrier(0);          // ... and older than this step's register fetches (counted wait at the end)
        // step s + 2's activations into the set step s's came from (split during step s - 1): a whole step to land, not the
        // few MFMAs left when the fetch waited for step s + 1's registers to be free (the latency was exposed on every step)
        if (FETCH_A) load_a(s + 2, rset[P]);
        __builtin_amdgcn_sched_barrier(0);
        h16x8 a[NPL][TM], b[NPL][TN];
        auto rd_a = [&](int q) __attribute__((always_inline)) {
#pragma unroll
            for (int i = 0; i < TM; ++i) a[q][i] = *reinterpret_cast<const h16x8*>(cur + q * XA_PLANE + fa + i * 32 * 8);
        };
        auto rd_b = [&](int q) __attribute__((always_inline)) {
#pragma unroll
            for (int i = 0; i < TN; ++i) b[q][i] = *reinterpret_cast<const h16x8*>(cur + q * B_PLANE + fb + i * 32 * 8);
        };
        rd_a(1); rd_b(0); rd_a(0); rd_b(1);                             // in the order the products below consume them
        // Hand-interleaved issue order (fenced so that the scheduler keeps it): after every MFMA two (TN = 1: four) of the 12 TM VALU instructions that
        // split step s + 1's activations (an MFMA holds the vector issue port for 8 of its 32 cycles), the plane writes as soon as a
        // row's planes are complete, the fetch of step s + 2 when the registers are free.  Low-order products first, a0·b0 last.
        float x[4 * TM];
        unsigned pk[2 * TM][NPL];
        float hi[2 * TM][2];
        if (FETCH_W) {
#pragma unroll
            // (v_med3: a value above its window's bound - a violated bound - saturates at fp16's largest finite value instead of splitting into inf - inf = NaN)
            for (int e = 0; e < 4 * TM; ++e) { x[e] = rset[P ^ 1][e >> 2][e & 3]; if (LRELU) x[e] = fmaxf(x[e], 0.2f * x[e]); x[e] = __builtin_amdgcn_fmed3f(x[e] * sa[e >> 2], -65504.f, 65504.f); }
        }
        auto split_op = [&](int k) __attribute__((always_inline)) {     // op k of 12 TM: pair k / 6 (two values), step k % 6
            const int pr = k / 6, o = k % 6;
            float& x0 = x[2 * pr]; float& x1 = x[2 * pr + 1];
            switch (o) {
                case 0: pk[pr][0] = cvt_pk_f16(x0, x1); break;
                case 1: hi[pr][0] = f16_lo(pk[pr][0]); break;
                case 2: hi[pr][1] = f16_hi(pk[pr][0]); break;
                case 3: x0 -= hi[pr][0]; break;
                case 4: x1 -= hi[pr][1]; break;
                default: pk[pr][1] = cvt_pk_f16(x0, x1); break;
            }
        };
        auto write_row = [&](int i) __attribute__((always_inline)) {
#pragma unroll
            for (int q = 0; q < NPL; ++q) {
                const u32x2 v = {pk[2 * i][q], pk[2 * i + 1][q]};
                *reinterpret_cast<u32x2*>(nxt + q * XA_PLANE + a_wr + i * 64 * 8) = v;
            }
        };
        __builtin_amdgcn_sched_barrier(0);
        constexpr int OPM = 4 / TN;                      // split instructions per MFMA: 12 TM over 3 TM TN
#pragma unroll
        for (int m = 0; m < 3 * TM * TN; ++m) {
            const int pr = m / (TM * TN), pa = F16_PA[pr], pb = F16_PB[pr], i = (m % (TM * TN)) / TN, j = m % TN;
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b[pb][j], a[pa][i], acc[i][j], 0, 0, 0);       // C^T tile
            if (FETCH_W) {
#pragma unroll
                for (int k = OPM * m; k < OPM * m + OPM; ++k) split_op(k);
                if ((OPM * (m + 1)) % 12 == 0) write_row((OPM * (m + 1)) / 12 - 1);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        // the weights of step s + 1 have landed and this wave's plane writes are done; step s + 2's activations stay in flight
        // (__syncthreads() would drain them: its fence waits for vmcnt(0))
        if (FETCH_A) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(TM) : "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    };
    using T = std::true_type; using F = std::false_type;
    using P0 = std::integral_constant<int, 0>; using P1 = std::integral_constant<int, 1>;
    for (int s = 0; s + 2 < nsteps; s += 2) {           // an even number of steps (gemm_h2_supports)
        step(s, P0{}, T{}, T{});
        step(s + 1, P1{}, T{}, T{});
    }
    step(nsteps - 2, P0{}, T{}, F{});
    step(nsteps - 1, P1{}, F{}, F{});

    // ---- epilogue: as in gemm_f32.hip (acc[i][j] = C^T of MFMA tile (i, j): lane & 31 = row, regs 4g..4g+3 = 4 columns)
    const bool vec_ok = ((p.ldc & 3) == 0) && (!p.residual || (p.ldr & 3) == 0) && ((p.N & 3) == 0);
    float* Cz = p.C + (size_t)blockIdx.z * p.slab_stride;
    const __amdgpu_buffer_rsrc_t rsC = make_rsrc(Cz + (size_t)m0 * p.ldc + n0);
    const __amdgpu_buffer_rsrc_t rsBias = make_rsrc(p.bias ? p.bias + n0 : p.A);
    const __amdgpu_buffer_rsrc_t rsWinv = make_rsrc(p.w_inv + n0);
    const __amdgpu_buffer_rsrc_t rsRb = make_rsrc(p.rowbias ? p.rowbias + n0 : p.A);
    const __amdgpu_buffer_rsrc_t rsRes = make_rsrc(p.residual ? p.residual + (size_t)m0 * p.ldr + n0 : p.A);

    if (vec_ok && n0 + TILE_N <= p.N) {
        constexpr int LDP = TILE_N + 4;
        constexpr int C4 = TILE_N / 4;
        static_assert(64 * LDP * 4 <= 2 * STAGE * 2, "epilogue staging fits the operand stages");
        float* stage = reinterpret_cast<float*>(h2_sm);
        // epilogue operands are fetched ahead of their use and ahead of the stores before them (mocha_gemm_h2p's epilogue has the reasoning):
        // the bias quad once per tile, the residual (else the row-bias) rows two store iterations ahead
        static_assert(256 % C4 == 0, "a thread keeps its column quad over the iterations");
        constexpr int NIT = 64 * C4 / 256, RSTEP = 256 / C4;
        const int c4 = tid % C4, r0 = tid / C4;
        const unsigned cb = (unsigned)c4 * 16u;
        const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
        const f32x4 bias4 = p.bias ? bload(rsBias, cb, 0u) : zero4;
        const f32x4 winv4 = bload(rsWinv, cb, 0u);                             // 1 / S_w[n] of this thread's column quad
        float mw[3] = {0.f, 0.f, 0.f};                                         // this thread's largest stored magnitude per window the tile touches
        const bool pre_res = p.residual != nullptr, pre_rb = !pre_res && p.rowbias != nullptr;
        f32x4 pre[2];
        auto fetch_pre = [&](int k) __attribute__((always_inline)) {
            const int rloc = 64 * (k / NIT) + r0 + RSTEP * (k % NIT);
            int row = m0 + rloc;
            row = row < p.M ? row : p.M - 1;                                          // rows past M: any valid address, the value is not used
            if (pre_res) pre[k & 1] = bload(rsRes, (unsigned)(row - m0) * (unsigned)p.ldr * 4u + cb, 0u);
            else if (pre_rb) pre[k & 1] = bload(rsRb, (unsigned)(row % p.rb_mod) * (unsigned)p.N * 4u + cb, 0u);
        };
        if (pre_res || pre_rb) { fetch_pre(0); fetch_pre(1); }
        const bool plain_out = !p.bias && !p.rowbias && !p.residual;          // nothing to fetch: read the stage and store (A/B: the pipelined form below costs such launches 5 %)
#pragma unroll
        for (int h = 0; h < TM; ++h) {                // 64 rows of the tile per pass
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int rblk = wm * TM + i;           // this wave's 32-row block of the tile
                if ((rblk >> 1) != h) continue;         // wave-uniform
                float* srow = stage + ((rblk & 1) * 32 + l31) * LDP + wn * (32 * TN) + 4 * hh;
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        f32x4 v = {acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
                        *reinterpret_cast<f32x4*>(srow + j * 32 + 8 * g) = v;
                    }
            }
            __syncthreads();
            if (plain_out) {
#pragma unroll
                for (int it = 0; it < NIT; ++it) {
                    const int r = r0 + RSTEP * it;
                    const int rloc = 64 * h + r;
                    if (m0 + rloc < p.M) {
                        f32x4 v = *reinterpret_cast<const f32x4*>(stage + r * LDP + c4 * 4) * (winv4 * h2_rowinv[rloc]);
                        if (p.act == 1) { v = mocha_gelu4(v); }
                        else if (p.act == 2) { v[0] = h2_lrelu(v[0]); v[1] = h2_lrelu(v[1]); v[2] = h2_lrelu(v[2]); v[3] = h2_lrelu(v[3]); }
                        else if (p.act == 3) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f); }
                        bstore(rsC, v, (unsigned)rloc * (unsigned)p.ldc * 4u + cb, 0u);
                        h2_track(mw, (rloc >= wb1) + (rloc >= wb2), fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
                    }
                }
            } else
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int r = r0 + RSTEP * it;
                const int rloc = 64 * h + r;
                f32x4 v = *reinterpret_cast<const f32x4*>(stage + r * LDP + c4 * 4) * (winv4 * h2_rowinv[rloc]) + bias4;
                if (pre_rb) v += pre[(NIT * h + it) & 1];
                else if (p.rowbias) v += bload(rsRb, (unsigned)((m0 + rloc) % p.rb_mod) * (unsigned)p.N * 4u + cb, 0u);   // a residual too: inline (2 GiB window)
                if (p.act == 1) { v = mocha_gelu4(v); }
                else if (p.act == 2) { v[0] = h2_lrelu(v[0]); v[1] = h2_lrelu(v[1]); v[2] = h2_lrelu(v[2]); v[3] = h2_lrelu(v[3]); }
                else if (p.act == 3) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f); }
                if (pre_res) v += pre[(NIT * h + it) & 1];
                if ((pre_res || pre_rb) && NIT * h + it + 2 < NIT * TM) fetch_pre(NIT * h + it + 2);      // ahead of this store
                if (m0 + rloc < p.M) {
                    bstore(rsC, v, (unsigned)rloc * (unsigned)p.ldc * 4u + cb, 0u);
                    h2_track(mw, (rloc >= wb1) + (rloc >= wb2), fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
                }
            }
            if (h + 1 < TM) __syncthreads();
        }
        h2_amax_out(p.c_amax, mw, h2_wmax, w0, nwin, lane, tid);
        return;
    }

    // ragged tiles (N not a multiple of 128, unaligned leading dimensions): straight from the accumulators
    float mwr[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int rloc = (wm * TM + i) * 32 + l31;
        const int row = m0 + rloc;
        if (row >= p.M) continue;
        const float inv_a = h2_rowinv[rloc];
        float rmax = 0.f;
        const float* rbrow = p.rowbias ? p.rowbias + (size_t)(row % p.rb_mod) * p.N : nullptr;
        const float* rsrow = p.residual ? p.residual + (size_t)row * p.ldr : nullptr;
        float* crow = Cz + (size_t)row * p.ldc;
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int c1 = n0 + wn * (32 * TN) + j * 32 + 8 * g + 4 * hh + e;
                    if (c1 >= p.N) continue;
                    float x = acc[i][j][4 * g + e] * (inv_a * p.w_inv[c1]);
                    if (p.bias) x += p.bias[c1];
                    if (rbrow) x += rbrow[c1];
                    if (p.act == 1) x = h2_gelu(x);
                    else if (p.act == 2) x = h2_lrelu(x);
                    else if (p.act == 3) x = fmaxf(x, 0.f);
                    if (rsrow) x += rsrow[c1];
                    crow[c1] = x;
                    rmax = fmaxf(rmax, fabsf(x));
                }
        h2_track(mwr, (rloc >= wb1) + (rloc >= wb2), rmax);
    }
    h2_amax_out(p.c_amax, mwr, h2_wmax, w0, nwin, lane, tid);
}


template <int TN> static constexpr size_t h2_lds_bytes() { return (size_t)2 * XT<TN>::STAGE * sizeof(unsigned short); }

template <bool L, bool G, int TM, int TN>
static hipError_t h2_attr() {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(&mocha_gemm_h2<L, G, TM, TN>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)h2_lds_bytes<TN>());
}

hipError_t gemm_h2_init() {
    hipError_t e = h2_attr<false, false, 2, 2>();
    if (e == hipSuccess) e = h2_attr<true, false, 2, 2>();
    if (e == hipSuccess) e = h2_attr<false, true, 2, 2>();
    if (e == hipSuccess) e = h2_attr<true, true, 2, 2>();
    if (e == hipSuccess) e = h2_attr<false, false, 1, 2>();
    if (e == hipSuccess) e = h2_attr<true, false, 1, 2>();
    if (e == hipSuccess) e = h2_attr<false, true, 1, 2>();
    if (e == hipSuccess) e = h2_attr<true, true, 1, 2>();
    if (e == hipSuccess) e = h2_attr<false, false, 2, 1>();
    if (e == hipSuccess) e = h2_attr<true, false, 2, 1>();
    if (e == hipSuccess) e = h2_attr<false, true, 2, 1>();
    if (e == hipSuccess) e = h2_attr<true, true, 2, 1>();
    if (e == hipSuccess) e = h2_attr<false, false, 1, 1>();
    if (e == hipSuccess) e = h2_attr<true, false, 1, 1>();
    if (e == hipSuccess) e = h2_attr<false, true, 1, 1>();
    if (e == hipSuccess) e = h2_attr<true, true, 1, 1>();
    return e;
}

// mid-size launches whose width is a multiple of 64 but not of 128 (to_mot's joint block, N = 192) take the 64 x 64 tile
static bool h2_tile64(const GemmParams& p) { return p.N % 64 == 0 && p.N % XN != 0 && gemm_is_small(p); }

// shapes this engine takes; everything else stays on the exact-f32 kernels
bool gemm_h2_supports(const GemmParams& p) {
    if (p.wsub || p.ksplit > 1) return false;       // the matcher's K-split contraction stays on the bf16 planes
    if (p.K % (2 * XK) != 0) return false;          // an even number of K steps (the register sets alternate)
    if (p.gather && (p.R != 1 || p.Cc % XK != 0)) return false;
    if (p.N % 64 != 0) return false;
    if (gemm_is_skinny(p)) return false;            // a handful of windows: latency-bound, mocha_gemm_skinny
    if (p.rows_per_win < 64) return false;          // a 128-row tile touches at most three windows
    return true;
}

template <int TM, int TN>
static void h2_launch(const GemmParams& p, hipStream_t s) {
    const int m_tiles = (p.M + TM * 64 - 1) / (TM * 64);
    const int m_pad = m_tiles >= 8 ? (m_tiles + 7) / 8 * 8 : m_tiles;
    const dim3 grid(m_pad * ((p.N + 64 * TN - 1) / (64 * TN)), 1, p.ksplit > 1 ? p.ksplit : 1);
    if (p.a_lrelu) {
        if (p.gather) hipLaunchKernelGGL((mocha_gemm_h2<true, true, TM, TN>), grid, dim3(256), h2_lds_bytes<TN>(), s, p);
        else hipLaunchKernelGGL((mocha_gemm_h2<true, false, TM, TN>), grid, dim3(256), h2_lds_bytes<TN>(), s, p);
    } else {
        if (p.gather) hipLaunchKernelGGL((mocha_gemm_h2<false, true, TM, TN>), grid, dim3(256), h2_lds_bytes<TN>(), s, p);
        else hipLaunchKernelGGL((mocha_gemm_h2<false, false, TM, TN>), grid, dim3(256), h2_lds_bytes<TN>(), s, p);
    }
}

hipError_t launch_gemm_h2(const GemmParams& p, hipStream_t s) {
    if (p.M <= 0 || p.N <= 0) return hipSuccess;
    if (!p.Wh2 || !p.w_inv || !p.a_amax || !gemm_h2_supports(p)) return hipErrorInvalidValue;
    if (p.gather && (long long)p.M / p.T_out * p.T_src * p.lda * 4 >= (1ll << 31)) return hipErrorInvalidValue;
    if (128ll * p.lda * 4 >= (1ll << 31)) return hipErrorInvalidValue;
    if (h2_tile64(p)) h2_launch<1, 1>(p, s);
    else if (p.ksplit <= 1 && p.N % XN != 0) h2_launch<2, 1>(p, s);
    else if (p.ksplit <= 1 && gemm_is_small(p)) h2_launch<1, 2>(p, s);
    else h2_launch<2, 2>(p, s);
    return hipGetLastError();
}

}  // namespace mocha
