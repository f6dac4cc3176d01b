// EXPERIMENT (round 5, not in the library): the two-plane fp16 / three-pass plane GEMM with a K step of 32 instead of 16 - one barrier and one
// LDS round trip per 24 MFMAs instead of per 12 (rocprofv3: the 16-deep kernel keeps the matrix pipe busy 0.31 of the time; its step's fixed
// part dominates).  A stage holds two 16-deep blocks [kb][plane][k half][row][8]; 68 KB of LDS for two stages: two workgroups per CU, 256 registers.
// 128 x 128 tiles, plain rows only (no gather, no LeakyReLU prologue, N % 128 == 0, K % 64 == 0); fixed scales as gemm_f16x2_r05.hip; through
// tools/gemm_bench only (tools/build_gemm_bench.sh builds gemm_bench_f16k32 from it).
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
#ifndef X3_SA_LOG2
#define X3_SA_LOG2 14
#endif
#ifndef X3_SW_LOG2
#define X3_SW_LOG2 14
#endif
static constexpr int NPL = 2;                            // planes per operand
static constexpr float X3_SA = (float)(1u << X3_SA_LOG2), X3_SW = (float)(1u << X3_SW_LOG2), X3_INV = 1.f / ((float)(1u << X3_SA_LOG2) * (float)(1u << X3_SW_LOG2));
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

#ifndef X3_MAXSUM
#define X3_MAXSUM 2         // products a_i b_j with i + j <= 2; tools/ builds an ablation with fewer (wrong results, timing only)
#endif
static constexpr int XN = 128, XK = 16;                  // tile width, K step; tile height = 64 TM rows
static constexpr int XA_HALF = 128 * 8 + 32;            // bf16 per k half of an A plane (2 KB + 64 B)
static constexpr int XA_PLANE = 2 * XA_HALF;            // 2176 bf16
static constexpr int XB_PLANE = 128 * 16;               // 2048 bf16, [k half][row][8]
static constexpr int XB_OFF = NPL * XA_PLANE;             // B planes follow the A planes of a stage
static constexpr int X_STAGE = XB_OFF + NPL * XA_PLANE;   // 13 056 bf16 = 26 112 B (the B planes use the padded A layout in LDS)
static constexpr int XW_BLOCK = NPL * XB_PLANE;           // packed weights per (n tile, k step): 6144 bf16 = 12 KB
// tile width 64 TN: the B planes of a stage hold 64 TN rows per k half (TN = 2: the layout above; TN = 1: one 64-row half of a packed block)
template <int TN> struct XT {
    static constexpr int TILE_N = 64 * TN;
    static constexpr int B_HALF = TILE_N * 8 + 32;
    static constexpr int B_PLANE = 2 * B_HALF;
    static constexpr int STAGE = XB_OFF + NPL * B_PLANE;
};
static_assert(XT<2>::STAGE == X_STAGE && XT<2>::B_HALF == XA_HALF, "TN = 2 is the 128-wide layout");

__device__ __forceinline__ float x3_lrelu(float x) { return x > 0.f ? x : 0.2f * x; }
__device__ __forceinline__ float x3_gelu(float x) { return mocha_gelu(x); }

// W [N][K] fp32 -> packed planes.  One workgroup per (n tile, k step) block: thread = (row, k half) reads 32 bytes and writes one
// 16-byte piece per plane, so every wave writes 512-byte runs of the 12 KB block (the image is written once per weight, but once per
// call for the matcher's transient bank).  wsub (K values, may be null) is subtracted from every row first (centred bank).
__global__ __launch_bounds__(256) void mocha_pack_x3(const float* __restrict__ W, const float* __restrict__ wsub, int N, int K, unsigned short* __restrict__ out) {
    const int ksteps = K / XK;
    const int nt = blockIdx.x / ksteps, ks = blockIdx.x - nt * ksteps;
    const int r = threadIdx.x >> 1, h = threadIdx.x & 1;
    const int n = nt * XN + r;
    const int k = ks * XK + 8 * h;
    f32x4 lo = {0.f, 0.f, 0.f, 0.f}, hi = lo;
    if (n < N) {
        lo = *reinterpret_cast<const f32x4*>(W + (size_t)n * K + k);
        hi = *reinterpret_cast<const f32x4*>(W + (size_t)n * K + k + 4);
        if (wsub) {
            lo -= *reinterpret_cast<const f32x4*>(wsub + k);
            hi -= *reinterpret_cast<const f32x4*>(wsub + k + 4);
        }
    }
    u32x2 a[NPL], b[NPL];
    f16_split4(lo * X3_SW, a); f16_split4(hi * X3_SW, b);
    unsigned short* blk = out + (size_t)blockIdx.x * XW_BLOCK;
#pragma unroll
    for (int q = 0; q < NPL; ++q) {
        const u32x4 v = {a[q][0], a[q][1], b[q][0], b[q][1]};
        *reinterpret_cast<u32x4*>(blk + q * XB_PLANE + h * 1024 + r * 8) = v;
    }
}

size_t gemm_x3_packed_elems(int N, int K) { return (size_t)((N + XN - 1) / XN) * (K / XK) * XW_BLOCK; }

hipError_t launch_pack_x3(const float* W, int N, int K, unsigned short* out, hipStream_t s, const float* wsub) {
    if (K % XK != 0) return hipErrorInvalidValue;
    const long long blocks = (long long)((N + XN - 1) / XN) * (K / XK);
    if (blocks > 0x7fffffffll) return hipErrorInvalidValue;
    hipLaunchKernelGGL(mocha_pack_x3, dim3((unsigned)blocks), dim3(256), 0, s, W, wsub, N, K, out);
    return hipGetLastError();
}

// LRELU: LeakyReLU(0.2) on the activations as they are split; GATHER: temporal-conv gather (kernels.h) instead of plain rows.
// Compile-time so that a K step is one basic block the scheduler can interleave.
// TM: 32-row MFMA blocks per wave: 2 = the 128-row tile; 1 = a 64-row tile (four waves of 32 x 64) for mid-size launches
// (a few dozen to a few hundred windows), where 128-row tiles would leave most workgroup slots empty.
// TN: 32-column MFMA blocks per wave: 2 = the 128-wide tile; 1 = a 128 x 64 tile (waves of 64 x 32) for N = 64 / 192 (to_mot's joint
// block), where a padded 128-wide tile would idle half the pipe.  (A 128 x 256 tile, TN = 4, was measured and not kept:
// tools/experiments/gemm_x3_tile_128x256.patch.txt.)
template <bool LRELU, bool GATHER, int TM, int TN>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void mocha_gemm_x3(GemmParams p) {
    constexpr int TILE_M = TM * 64;
    constexpr int TILE_N = XT<TN>::TILE_N, B_HALF = XT<TN>::B_HALF, B_PLANE = XT<TN>::B_PLANE, SUB = XT<TN>::STAGE, STAGE = 2 * SUB;      // a stage = two 16-deep blocks
    static_assert(TM == 2 && TN == 2 && !GATHER && !LRELU, "prototype: 128 x 128 tiles on plain rows");
    extern __shared__ __attribute__((aligned(16))) unsigned short x3_sm[];          // [2][STAGE]
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hh = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;

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
    const int steps_total = p.K / (2 * XK);               // 32-deep steps
    const int per = (steps_total + p.ksplit - 1) / p.ksplit;
    const int s0 = blockIdx.z * per;
    const int nsteps = (s0 + per) <= steps_total ? per : steps_total - s0;
#ifdef X3_EXP_STAMPS       // diagnostic build (tools/): cycle stamps per workgroup into the buffer passed as p.wsub
    long long stamp[4];
    stamp[3] = (long long)__builtin_amdgcn_s_memrealtime();      // 100 MHz, common to the chip
    stamp[0] = (long long)__builtin_amdgcn_s_memtime();
#define X3_STAMP(i) stamp[i] = (long long)__builtin_amdgcn_s_memtime()
#define X3_STAMP_OUT() do { if (tid == 0) { long long* d = (long long*)p.wsub + (size_t)bid * 6; d[0] = stamp[0]; d[1] = stamp[1]; d[2] = stamp[2]; d[3] = (long long)__builtin_amdgcn_s_memtime(); d[4] = stamp[3]; d[5] = (long long)__builtin_amdgcn_s_memrealtime(); } } while (0)
#else
#define X3_STAMP(i)
#define X3_STAMP_OUT()
#endif

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
    f32x4 rset[2][2][TM];                           // [set][16-deep block][row]: step t's activations wait in set t & 1, fetched two steps ahead
    auto load_a = [&](int s, f32x4 (&ra)[2][TM]) __attribute__((always_inline)) {
        const int k0 = (s0 + s) * 2 * XK;
        if (!GATHER) {
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int i = 0; i < TM; ++i) ra[kb][i] = bload(rsA, a_off[i], (unsigned)(k0 + XK * kb) * 4u);
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
                ra[0][i] = bload(rsA, off, (unsigned)cin * 4u);
            }
        }
    };
    // plane q of (row, piece lc): k half lc >> 1, 8 bytes at (lc & 1)
    const int a_wr = (lc >> 1) * XA_HALF + lrow * 8 + (lc & 1) * 4;
    auto split_store = [&](const f32x4 (&ra)[2][TM], unsigned short* st) __attribute__((always_inline)) {
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            f32x4 v = ra[kb][i];
            if (LRELU) { v[0] = x3_lrelu(v[0]); v[1] = x3_lrelu(v[1]); v[2] = x3_lrelu(v[2]); v[3] = x3_lrelu(v[3]); }
            u32x2 pl[NPL];
            f16_split4(v * X3_SA, pl);
#pragma unroll
            for (int q = 0; q < NPL; ++q) *reinterpret_cast<u32x2*>(st + kb * SUB + q * XA_PLANE + a_wr + i * 64 * 8) = pl[q];
        }
    };
    // ---- W: linear copy of the packed 12 KB block of (nt, step) into the stage
    const int wblock = TN == 1 ? nt >> 1 : nt;           // the packed image is in 128-column blocks; a 64-wide tile takes one half of one
    const __amdgpu_buffer_rsrc_t rsW = make_rsrc(p.Wsplit + ((size_t)wblock * steps_total + s0) * 2 * XW_BLOCK);
    auto dma_w = [&](int s, unsigned short* st) __attribute__((always_inline)) {
        // sixteen 1 KB pieces per thread-quartet: piece pc = j * 4 + wave = (block kb, plane, k half, 64-row half) of the two packed 8 KB blocks
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int pc = j * 4 + wave;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (__attribute__((address_space(3))) void*)(st + (pc >> 3) * SUB + XB_OFF + ((pc >> 2) & 1) * B_PLANE +
                                                     ((pc >> 1) & 1) * B_HALF + (pc & 1) * 512), 16,
                                                     (unsigned)(j * 256 + tid) * 16u, (unsigned)s * (2u * XW_BLOCK * 2u), 0, 0);
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
    dma_w(0, x3_sm);
    __builtin_amdgcn_sched_barrier(0);
    split_store(rset[0], x3_sm);
    load_a(1, rset[1]);                             // K >= 32 (gemm_x3_supports)
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(2 * TM) : "memory");     // step 1's 2 TM fetches stay in flight

    // One K step.  FETCH_W: step s + 1 exists (its weights are copied and its activations split into the other stage);
    // FETCH_A: step s + 2 exists (its activations are fetched).  The three variants are straight-line code, so the compiler's own
    // vmcnt bookkeeping for the activation registers is exact: the split waits for the two oldest fetches only, not for the copy.
    auto step = [&](int s, auto parity, auto fetch_w, auto fetch_a) __attribute__((always_inline)) {
        constexpr bool FETCH_W = decltype(fetch_w)::value, FETCH_A = decltype(fetch_a)::value;
        constexpr int P = decltype(parity)::value;      // s & 1
        unsigned short* cur = x3_sm + P * STAGE;
        unsigned short* nxt = x3_sm + (P ^ 1) * STAGE;
        if (FETCH_W) dma_w(s + 1, nxt);             // first thing after the barrier: a whole step to land
        __builtin_amdgcn_sched_barrier(0);          // ... and older than this step's register fetches (counted wait at the end)
        // step s + 2's activations into the set step s's came from (split during step s - 1): a whole step to land, not the
        // few MFMAs left when the fetch waited for step s + 1's registers to be free (the latency was exposed on every step)
        if (FETCH_A) load_a(s + 2, rset[P]);
        __builtin_amdgcn_sched_barrier(0);
        h16x8 a[2][NPL][TM], b[2][NPL][TN];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {                 // in the order the products below consume them
#pragma unroll
            for (int i = 0; i < TM; ++i) a[kb][1][i] = *reinterpret_cast<const h16x8*>(cur + kb * SUB + 1 * XA_PLANE + fa + i * 32 * 8);
#pragma unroll
            for (int i = 0; i < TN; ++i) b[kb][0][i] = *reinterpret_cast<const h16x8*>(cur + kb * SUB + 0 * B_PLANE + fb + i * 32 * 8);
#pragma unroll
            for (int i = 0; i < TM; ++i) a[kb][0][i] = *reinterpret_cast<const h16x8*>(cur + kb * SUB + 0 * XA_PLANE + fa + i * 32 * 8);
#pragma unroll
            for (int i = 0; i < TN; ++i) b[kb][1][i] = *reinterpret_cast<const h16x8*>(cur + kb * SUB + 1 * B_PLANE + fb + i * 32 * 8);
        }
        // 48 split instructions of step s + 1's activations (eight value pairs x six) over the 24 MFMAs, two per MFMA; a row's planes are written as
        // soon as they are complete
        float x[8 * TM];
        unsigned pk[4 * TM][NPL];
        float hi[4 * TM][2];
        if (FETCH_W) {
#pragma unroll
            for (int e = 0; e < 8 * TM; ++e) { x[e] = rset[P ^ 1][e / (4 * TM)][(e / 4) % TM][e & 3] * X3_SA; }
        }
        auto split_op = [&](int k) __attribute__((always_inline)) {     // op k of 48: pair k / 6 (two values), step k % 6
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
        auto write_row = [&](int r) __attribute__((always_inline)) {    // r = kb * TM + i: pairs 2 r, 2 r + 1
            const int kb = r / TM, i = r % TM;
#pragma unroll
            for (int q = 0; q < NPL; ++q) {
                const u32x2 v = {pk[2 * r][q], pk[2 * r + 1][q]};
                *reinterpret_cast<u32x2*>(nxt + kb * SUB + q * XA_PLANE + a_wr + i * 64 * 8) = v;
            }
        };
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = 0; m < 2 * 3 * TM * TN; ++m) {
            const int kb = m / (3 * TM * TN), mm = m % (3 * TM * TN);
            const int pr = mm / (TM * TN), pa = F16_PA[pr], pb = F16_PB[pr], i = (mm % (TM * TN)) / TN, j = mm % TN;
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b[kb][pb][j], a[kb][pa][i], acc[i][j], 0, 0, 0);       // C^T tile
            if (FETCH_W) {
                split_op(2 * m); split_op(2 * m + 1);
                if ((2 * (m + 1)) % 12 == 0) write_row((2 * (m + 1)) / 12 - 1);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (FETCH_A) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(2 * TM) : "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    };
    X3_STAMP(1);
    using T = std::true_type; using F = std::false_type;
    using P0 = std::integral_constant<int, 0>; using P1 = std::integral_constant<int, 1>;
    for (int s = 0; s + 2 < nsteps; s += 2) {           // an even number of steps (gemm_x3_supports)
        step(s, P0{}, T{}, T{});
        step(s + 1, P1{}, T{}, T{});
    }
    step(nsteps - 2, P0{}, T{}, F{});
    step(nsteps - 1, P1{}, F{}, F{});
    X3_STAMP(2);

    // ---- epilogue: as in gemm_f32.hip (acc[i][j] = C^T of MFMA tile (i, j): lane & 31 = row, regs 4g..4g+3 = 4 columns)
    const bool vec_ok = ((p.ldc & 3) == 0) && (!p.residual || (p.ldr & 3) == 0) && ((p.N & 3) == 0);
    float* Cz = p.C + (size_t)blockIdx.z * p.slab_stride;
    const __amdgpu_buffer_rsrc_t rsC = make_rsrc(Cz + (size_t)m0 * p.ldc + n0);
    const __amdgpu_buffer_rsrc_t rsBias = make_rsrc(p.bias ? p.bias + n0 : p.A);
    const __amdgpu_buffer_rsrc_t rsRb = make_rsrc(p.rowbias ? p.rowbias + n0 : p.A);
    const __amdgpu_buffer_rsrc_t rsRes = make_rsrc(p.residual ? p.residual + (size_t)m0 * p.ldr + n0 : p.A);

    if (vec_ok && n0 + TILE_N <= p.N) {
        constexpr int LDP = TILE_N + 4;
        constexpr int C4 = TILE_N / 4;
        static_assert(64 * LDP * 4 <= 2 * STAGE * 2, "epilogue staging fits the operand stages");
        float* stage = reinterpret_cast<float*>(x3_sm);
        // epilogue operands are fetched ahead of their use and ahead of the stores before them (mocha_gemm_x3p's epilogue has the reasoning):
        // the bias quad once per tile, the residual (else the row-bias) rows two store iterations ahead
        static_assert(256 % C4 == 0, "a thread keeps its column quad over the iterations");
        constexpr int NIT = 64 * C4 / 256, RSTEP = 256 / C4;
        const int c4 = tid % C4, r0 = tid / C4;
        const unsigned cb = (unsigned)c4 * 16u;
        const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
        const f32x4 bias4 = p.bias ? bload(rsBias, cb, 0u) : zero4;
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
                        f32x4 v = *reinterpret_cast<const f32x4*>(stage + r * LDP + c4 * 4) * X3_INV;
                        if (p.act == 1) { v = mocha_gelu4(v); }
                        else if (p.act == 2) { v[0] = x3_lrelu(v[0]); v[1] = x3_lrelu(v[1]); v[2] = x3_lrelu(v[2]); v[3] = x3_lrelu(v[3]); }
                        else if (p.act == 3) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f); }
                        bstore(rsC, v, (unsigned)rloc * (unsigned)p.ldc * 4u + cb, 0u);
                    }
                }
            } else
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int r = r0 + RSTEP * it;
                const int rloc = 64 * h + r;
                f32x4 v = *reinterpret_cast<const f32x4*>(stage + r * LDP + c4 * 4) * X3_INV + bias4;
                if (pre_rb) v += pre[(NIT * h + it) & 1];
                else if (p.rowbias) v += bload(rsRb, (unsigned)((m0 + rloc) % p.rb_mod) * (unsigned)p.N * 4u + cb, 0u);   // a residual too: inline (2 GiB window)
                if (p.act == 1) { v = mocha_gelu4(v); }
                else if (p.act == 2) { v[0] = x3_lrelu(v[0]); v[1] = x3_lrelu(v[1]); v[2] = x3_lrelu(v[2]); v[3] = x3_lrelu(v[3]); }
                else if (p.act == 3) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f); }
                if (pre_res) v += pre[(NIT * h + it) & 1];
                if ((pre_res || pre_rb) && NIT * h + it + 2 < NIT * TM) fetch_pre(NIT * h + it + 2);      // ahead of this store
                if (m0 + rloc < p.M) bstore(rsC, v, (unsigned)rloc * (unsigned)p.ldc * 4u + cb, 0u);
            }
            if (h + 1 < TM) __syncthreads();
        }
        X3_STAMP_OUT();
        return;
    }

    // ragged tiles (N not a multiple of 128, unaligned leading dimensions): straight from the accumulators
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int row = m0 + (wm * TM + i) * 32 + l31;
        if (row >= p.M) continue;
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
                    float x = acc[i][j][4 * g + e] * X3_INV;
                    if (p.bias) x += p.bias[c1];
                    if (rbrow) x += rbrow[c1];
                    if (p.act == 1) x = x3_gelu(x);
                    else if (p.act == 2) x = x3_lrelu(x);
                    else if (p.act == 3) x = fmaxf(x, 0.f);
                    if (rsrow) x += rsrow[c1];
                    crow[c1] = x;
                }
    }
}


template <int TN> static constexpr size_t x3_lds_bytes() { return (size_t)4 * XT<TN>::STAGE * sizeof(unsigned short); }

template <bool L, bool G, int TM, int TN>
static hipError_t x3_attr() {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(&mocha_gemm_x3<L, G, TM, TN>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)x3_lds_bytes<TN>());
}

hipError_t gemm_x3_init() { return x3_attr<false, false, 2, 2>(); }

// The 64 x 64 tile (round 5): four waves of 32 x 32, six MFMAs per wave and K step.
//  * mid-size launches whose width is a multiple of 64 but not of 128 (to_mot's joint block, N = 192, at a few dozen to a few hundred
//    windows) take it ALWAYS: their alternative was the exact-f32 engine (128 windows: 18.7 -> 13.4 us);
//  * mid-size launches of 128-multiples take it when they have fewer 64 x 128 tiles than p.tile64_below (option "gemm_tile64_below",
//    default 0 = never): measured NO gain - 128 windows, N = 256: K = 512 26.8 -> 27.5 us, K = 1024 47.2 -> 49.6 us; the whole 128-window
//    step 1.116 -> 1.150 ms (profiles/r05/c_tile64.txt) - a mid-size launch's time is its tiles' serial K loop, and a K step's floor
//    (barrier, LDS round trip, the activation split) does not shrink with the tile.
static bool x3_tile64(const GemmParams& p) {
    if (p.ksplit > 1 || p.N % 64 != 0 || !gemm_is_small(p)) return false;
    if (p.N % XN != 0) return true;
    const long long t64x128 = (long long)((p.M + 63) / 64) * ((p.N + 127) / 128);
    return t64x128 < p.tile64_below;
}

// shapes this engine takes; everything else stays on the exact-f32 kernels
bool gemm_x3_supports(const GemmParams& p) {
    return !p.wsub && p.ksplit <= 1 && !p.gather && !p.a_lrelu && p.K % 64 == 0 && p.N % XN == 0 && !gemm_is_small(p) && !gemm_is_skinny(p);
}

hipError_t launch_gemm_x3(const GemmParams& p, hipStream_t s) {
    if (p.M <= 0 || p.N <= 0) return hipSuccess;
    if (!p.Wsplit || !gemm_x3_supports(p)) return hipErrorInvalidValue;
    const int m_tiles = (p.M + 127) / 128;
    const int m_pad = m_tiles >= 8 ? (m_tiles + 7) / 8 * 8 : m_tiles;
    hipLaunchKernelGGL((mocha_gemm_x3<false, false, 2, 2>), dim3(m_pad * (p.N / 128)), dim3(256), x3_lds_bytes<2>(), s, p);
    return hipGetLastError();
}

}  // namespace mocha
