// Context matching for MANY queries (batched characterization; BASELINE configs[2]/[3]: 128-1024 windows against a
// 4k-16k entry bank):  exact 1-NN = argmin_b sum_d (q_d - b_d)^2   (BallTree.query(k=1), test_fullframework.py:296,443).
//
//   coarse pass   S = (q - c)(b - c)^T on the matrix pipe, c = bank centroid (distances are translation invariant and the
//                 centred operands are of the size of the distances themselves, see mocha_api.cpp::do_match);
//                   fp32 bank: mocha_gemm_f32 (exact-f32 MFMA, gemm_f32.hip);
//                   bf16 bank: mocha_match_gemm_bf16 below - ONE bf16 plane of the centred query against the centred
//                   bf16 bank.  Q <= ~300 is HBM-bound (the bank streams through once: 2 N D bytes), so the kernel is
//                   built as a stream: the K range is split over the 8 XCDs (an XCD's workgroups share one K slice of the
//                   queries in its L2, and every bank byte is fetched exactly once), operands go global -> LDS by LDS-DMA
//                   through a 4-stage ring (96 KB in flight per CU), one barrier per step;
//   select        mocha_match_select: per query every row whose coarse score ||b-c||^2 - 2 S lies within the coarse pass's
//                 error bound of the best one is re-evaluated EXACTLY, sum ((q-c) - b)^2 in the direct form; the smallest
//                 wins, ties to the lowest index (see the kernel's header for the bounds).
#include "kernels.h"
#include "device_utils.h"
#include <cstdlib>

namespace mocha {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));


// out = bf16(x - centre) row by row: the centred queries of the bf16 coarse pass.  One thread per 8 elements, flat grid
// (a few hundred query rows do not fill the chip with one workgroup per row).
__global__ __launch_bounds__(256) void mocha_center_bf16(const float* __restrict__ x, const float* __restrict__ c,
                                                         unsigned short* __restrict__ out, int cols8, long long total8) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total8) return;
    const int ci = (int)(i % cols8);
    const f32x4* xr = reinterpret_cast<const f32x4*>(x) + 2 * i;
    const f32x4* cr = reinterpret_cast<const f32x4*>(c) + 2 * ci;
    const f32x4 a = xr[0] - cr[0], b = xr[1] - cr[1];
    u32x4 w;
    w[0] = bf16_bits(a[0]) | (bf16_bits(a[1]) << 16); w[1] = bf16_bits(a[2]) | (bf16_bits(a[3]) << 16);
    w[2] = bf16_bits(b[0]) | (bf16_bits(b[1]) << 16); w[3] = bf16_bits(b[2]) | (bf16_bits(b[3]) << 16);
    reinterpret_cast<u32x4*>(out)[i] = w;
}

hipError_t launch_center_bf16(const float* x, const float* centre, void* out, int64_t rows, int cols, hipStream_t s) {
    if (rows <= 0) return hipSuccess;
    if (cols % 8) return hipErrorInvalidValue;
    const long long total8 = (long long)rows * (cols / 8);
    hipLaunchKernelGGL(mocha_center_bf16, dim3((unsigned)((total8 + 255) / 256)), dim3(256), 0, s, x, centre, (unsigned short*)out,
                       cols / 8, total8);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------------
// S[z][q][n] = sum_{k in slice z} A[q][k] B[n][k],  A (Q, D) and B (N, D) bf16, k contiguous.
// 128 x 128 tile per 512-thread workgroup (8 waves of 64 x 32: two 32x32x16 MFMA tiles each), 64 k per step.
// ---------------------------------------------------------------------------------------------------------------------
struct MatchGemmParams {
    const unsigned short* A; const unsigned short* B; float* S;
    int Q; long long N; int D; int ksplit; long long slab_stride; int m_tiles, n_tiles;
    long long a_plane;            // elements between the two stacked query planes (NPL == 2)
    const unsigned short* Bt;     // the bank as the kernel's own LDS image, [n tile][k step][128 rows][64] with the swizzle applied (mocha_tile_bf16), or null
    // round 6 (option "match_fold"): arrival counters per (m tile, n tile), zero at first use and left at zero; non-null: the LAST of a tile's
    // K-slab workgroups adds the slabs up, in slab order, into slab 0 - the selection then reads one slab instead of `ksplit`
    unsigned* tickets;
};

static constexpr int MG_BM = 128, MG_BN = 128, MG_BK = 64;

// ---------------------------------------------------------------------------------------------------------------------
// Operands go global -> LDS directly (buffer_load ... lds, no VGPR staging, no ds_write), through a ring of
// R stages with R - 1 steps in flight; one barrier per step.  256 threads, each wave a 64 x 64 sub-tile (one LDS read per
// MFMA).  LDS rows are unpadded 128-byte rows (a DMA instruction writes 1 KB = 8 rows in lane order); the 16-byte piece c of
// row r lives in slot c ^ ((r >> 1) & 7), which makes the MFMA operand reads (ds_read_b128: 16-lane groups take rows of all 16
// residues mod 16 at one k piece) conflict-free; the swizzle is applied by choosing which global piece a lane fetches.
// ---------------------------------------------------------------------------------------------------------------------
// NPL = 2 (round 4): the queries come as TWO stacked bf16 planes, a = a0 + a1 to 16 significant bits, and S accumulates both products.  The
// pass is HBM-bound on the bank (a step's 36 - 48 KB arrive in ~0.8 us, its 16 / 32 MFMAs per wave take 0.25 / 0.5 us), so the second plane
// costs LDS (a stage grows from 32 to 48 KB: ring of 3) and 6 MB of query traffic, not time - and shrinks the selection's error bound, and
// with it the rows that must be re-evaluated exactly, by 2^8.
// NTB (round 5): the bank's LDS-DMA loads carry the non-temporal hint (aux = 2): read once - with caches full of other kernels' dirty lines
// (the state inside mocha_characterize) they do not wait for write-backs they do not need (tools/match_pass_probe.hip)
template <int R, int NPL, bool NTB>
__global__ __launch_bounds__(256) void mocha_match_gemm_bf16_dma(MatchGemmParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned short mg_sm[];          // [R][NPL * BM + BN][64]
    constexpr int STAGE = (NPL * MG_BM + MG_BN) * MG_BK;                             // bf16 per stage
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hh = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    const int x = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int mt = j % p.m_tiles, pl = j / p.m_tiles;
    const int pp = pl * 8 + x;
    const int z = pp % p.ksplit, nt = pp / p.ksplit;
    if (nt >= p.n_tiles) return;
    const int m0 = mt * MG_BM;
    const long long n0 = (long long)nt * MG_BN;
    const int steps_total = p.D / MG_BK;
    const int per = (steps_total + p.ksplit - 1) / p.ksplit;
    const int s_begin = z * per;
    const int s_end = (s_begin + per) < steps_total ? (s_begin + per) : steps_total;
    const int nsteps = s_end > s_begin ? s_end - s_begin : 0;

    const __amdgpu_buffer_rsrc_t rsA = make_rsrc(p.A + (size_t)m0 * p.D);
    // tiled bank image: a (tile, step) block is 16 KB of contiguous memory in LDS order - every DMA instruction copies 1 KB linearly and a
    // workgroup walks one contiguous 16 KB x steps region (row-major rows give 128-byte pieces 46 KB apart: 0.45 of the HBM peak when the
    // bank is not already in the Infinity Cache, measured inside characterize)
    const __amdgpu_buffer_rsrc_t rsB = p.Bt ? make_rsrc(p.Bt + (size_t)nt * (size_t)steps_total * (MG_BN * MG_BK)) : make_rsrc(p.B + (size_t)n0 * p.D);
    const __amdgpu_buffer_rsrc_t rsA1 = make_rsrc(p.A + (size_t)(NPL == 2 ? p.a_plane : 0) + (size_t)m0 * p.D);
    // DMA pieces: wave w fills 8-row pieces w, w + 4, w + 8, w + 12 of A and of B; lane -> (row in piece, slot)
    unsigned a_off[4], b_off[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = 8 * (wave + 4 * i) + (lane >> 3);             // row in the tile
        const int c = (lane & 7) ^ ((r >> 1) & 7);                  // global 16-byte piece that belongs in this lane's slot
        int ra = m0 + r; ra = ra < p.Q ? ra : p.Q - 1;
        long long rb = n0 + r; rb = rb < p.N ? rb : p.N - 1;
        a_off[i] = ((unsigned)(ra - m0) * (unsigned)p.D + c * 8u) * 2u;
        b_off[i] = p.Bt ? (unsigned)((wave + 4 * i) * 512 + lane * 8) * 2u : ((unsigned)(rb - n0) * (unsigned)p.D + c * 8u) * 2u;
    }
    auto issue = [&](int s) __attribute__((always_inline)) {        // step s (relative) -> ring slot s % R; past the end: re-fetch the last step into a dead slot
        const int sc = s < nsteps ? s : nsteps - 1;
        const unsigned so = (unsigned)((s_begin + sc) * MG_BK) * 2u;
        const unsigned sob = p.Bt ? (unsigned)(s_begin + sc) * (unsigned)(MG_BN * MG_BK * 2) : so;
        unsigned short* st = mg_sm + (s % R) * STAGE;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (__attribute__((address_space(3))) void*)(st + (wave + 4 * i) * 512), 16, a_off[i], so, 0, 0);
            if (NPL == 2)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA1, (__attribute__((address_space(3))) void*)(st + MG_BM * MG_BK + (wave + 4 * i) * 512), 16, a_off[i], so, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (__attribute__((address_space(3))) void*)(st + NPL * MG_BM * MG_BK + (wave + 4 * i) * 512), 16, b_off[i], sob, 0, NTB ? 2 : 0);
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][jj][r] = 0.f;

    // reader offsets (bf16 units inside a stage): row r, k piece c = 2 ks + hh -> slot c ^ ((r >> 1) & 7)
    unsigned ra_off[2], rb_off[2], key_a[2], key_b[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int r_a = wm * 64 + i * 32 + l31, r_b = wn * 64 + i * 32 + l31;
        ra_off[i] = (unsigned)r_a * 64u; key_a[i] = (unsigned)((r_a >> 1) & 7);
        rb_off[i] = (unsigned)(NPL * MG_BM + r_b) * 64u; key_b[i] = (unsigned)((r_b >> 1) & 7);
    }

    if (nsteps > 0) {
#pragma unroll
        for (int u = 0; u < R - 1; ++u) issue(u);
        for (int s = 0; s < nsteps; ++s) {
            // the 8 pieces of step s are this wave's oldest; (R - 2) younger steps may stay in flight
            asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"((R - 2) * 4 * (NPL + 1)) : "memory");
            issue(s + R - 1);                                       // into the slot every wave finished reading before the barrier
            const unsigned short* st = mg_sm + (s % R) * STAGE;
#pragma unroll
            for (int ks = 0; ks < MG_BK / 16; ++ks) {
                s16x8 a[2], a1[2], b[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    a[i] = *reinterpret_cast<const s16x8*>(st + ra_off[i] + (((unsigned)(2 * ks + hh) ^ key_a[i]) << 3));
                    if (NPL == 2) a1[i] = *reinterpret_cast<const s16x8*>(st + MG_BM * MG_BK + ra_off[i] + (((unsigned)(2 * ks + hh) ^ key_a[i]) << 3));
                    b[i] = *reinterpret_cast<const s16x8*>(st + rb_off[i] + (((unsigned)(2 * ks + hh) ^ key_b[i]) << 3));
                }
                // the low plane first: its products are 2^-8 of the high plane's
                if (NPL == 2) {
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int jj = 0; jj < 2; ++jj)
                            acc[i][jj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[jj], a1[i], acc[i][jj], 0, 0, 0);
                }
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int jj = 0; jj < 2; ++jj)
                        acc[i][jj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[jj], a[i], acc[i][jj], 0, 0, 0);    // C^T tile: lane = query row
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // the dead-slot fetches of the tail
    }

    float* Sz = p.S + (size_t)z * p.slab_stride;
    const bool vec = (p.N & 3) == 0;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = m0 + wm * 64 + i * 32 + l31;
        if (row >= p.Q) continue;
        float* srow = Sz + (size_t)row * p.N;
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const long long col = n0 + wn * 64 + jj * 32 + 8 * g + 4 * hh;
                if (col >= p.N) continue;
                if (vec) {
                    const f32x4 v = {acc[i][jj][4 * g], acc[i][jj][4 * g + 1], acc[i][jj][4 * g + 2], acc[i][jj][4 * g + 3]};
                    *reinterpret_cast<f32x4*>(srow + col) = v;
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (col + e < p.N) srow[col + e] = acc[i][jj][4 * g + e];
                }
            }
    }
    if (!p.tickets || p.ksplit <= 1) return;
    // ---- fold (round 6): the tile's slabs live on up to 8 XCDs; publish mine (agent-scope release), take a ticket, and the last arrival
    // reads all of them back (acquire) and leaves their sum - added in slab order from zero, exactly as mocha_match_select adds them - in slab 0
    __shared__ int s_last;
    __threadfence();
    __syncthreads();
    if (tid == 0) {
        const unsigned t = atomicAdd(p.tickets + (size_t)mt * p.n_tiles + nt, 1u);
        s_last = t == (unsigned)p.ksplit - 1u;
    }
    __syncthreads();
    if (!s_last) return;
    __threadfence();
    if (vec) {
        const int c4 = tid & 31, r8 = tid >> 5;                  // a 32-lane half covers one row's 128 columns; 8 rows per iteration
#pragma unroll 2
        for (int it = 0; it < MG_BM / 8; ++it) {
            const int row = m0 + it * 8 + r8;
            const long long col = n0 + 4 * c4;
            if (row < p.Q && col < p.N) {
                const float* src = p.S + (size_t)row * p.N + col;
                f32x4 part[8];
#pragma unroll
                for (int zz = 0; zz < 8; ++zz)
                    part[zz] = zz < p.ksplit ? __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(src + (size_t)zz * p.slab_stride)) : f32x4{0.f, 0.f, 0.f, 0.f};
                f32x4 sum = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int zz = 0; zz < 8; ++zz)
                    if (zz < p.ksplit) sum += part[zz];
                *reinterpret_cast<f32x4*>(p.S + (size_t)row * p.N + col) = sum;
            }
        }
    } else {
        for (int e = tid; e < MG_BM * MG_BN; e += 256) {
            const int row = m0 + e / MG_BN; const long long col = n0 + e % MG_BN;
            if (row < p.Q && col < p.N) {
                float sum = 0.f;
                for (int zz = 0; zz < p.ksplit; ++zz) sum += p.S[(size_t)zz * p.slab_stride + (size_t)row * p.N + col];
                p.S[(size_t)row * p.N + col] = sum;
            }
        }
    }
    if (tid == 0) p.tickets[(size_t)mt * p.n_tiles + nt] = 0u;      // for the next launch (same stream)
}

template <int R, int NPL>
static constexpr size_t mg_dma_lds_bytes() { return (size_t)R * (NPL * MG_BM + MG_BN) * MG_BK * sizeof(unsigned short); }

hipError_t match_select_init();

hipError_t match_mfma_init() {
    constexpr size_t lds2 = mg_dma_lds_bytes<3, 2>(), lds1 = mg_dma_lds_bytes<4, 1>();
    hipError_t r = hipFuncSetAttribute(reinterpret_cast<const void*>(&mocha_match_gemm_bf16_dma<4, 1, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1);
    if (r == hipSuccess) r = hipFuncSetAttribute(reinterpret_cast<const void*>(&mocha_match_gemm_bf16_dma<4, 1, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1);
    if (r == hipSuccess) r = hipFuncSetAttribute(reinterpret_cast<const void*>(&mocha_match_gemm_bf16_dma<3, 2, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2);
    if (r == hipSuccess) r = hipFuncSetAttribute(reinterpret_cast<const void*>(&mocha_match_gemm_bf16_dma<3, 2, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2);
    if (r == hipSuccess) r = match_select_init();
    return r;
}

// K split of the bf16 coarse pass: a power of two <= 8 (one K slice per XCD, or per group of XCDs) with about two
// workgroups per CU; fewer slices for large query sets, whose partial-score slabs would otherwise dominate the traffic
int match_bf16_ksplit(int Q, int64_t N) {
    const long long tiles = (long long)((Q + MG_BM - 1) / MG_BM) * ((N + MG_BN - 1) / MG_BN);
    int k = 8;
    while (k > 1 && tiles * k > 512 && tiles * (k / 2) >= 256) k >>= 1;
    return k;
}

// bank16 (N, D) bf16 row-major -> the coarse pass's own image: block (n tile, k step) = 128 rows x 64 k in LDS order, 16-byte piece c of row r
// at slot c ^ ((r >> 1) & 7); rows past N are zero.  One 256-thread workgroup per block.
__global__ __launch_bounds__(256) void mocha_tile_bf16(const unsigned short* __restrict__ bank16, unsigned short* __restrict__ out, long long N, int D) {
    const int steps = D / MG_BK;
    const int nt = blockIdx.x / steps, st = blockIdx.x - nt * steps;
    unsigned short* blk = out + (size_t)blockIdx.x * (MG_BN * MG_BK);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int pc = threadIdx.x + 256 * i;                  // piece of the block: row r, slot
        const int r = pc >> 3, slot = pc & 7;
        const int c = slot ^ ((r >> 1) & 7);
        const long long row = (long long)nt * MG_BN + r;
        u32x4 v = {0u, 0u, 0u, 0u};
        if (row < N) v = *reinterpret_cast<const u32x4*>(bank16 + (size_t)row * D + st * MG_BK + c * 8);
        *reinterpret_cast<u32x4*>(blk + pc * 8) = v;
    }
}

size_t match_tiled_elems(int64_t N, int D) { return (size_t)((N + MG_BN - 1) / MG_BN) * MG_BN * (size_t)D; }

hipError_t launch_tile_bf16(const void* bank16, void* out, int64_t N, int D, hipStream_t s) {
    if (N <= 0) return hipSuccess;
    if (D % MG_BK) return hipErrorInvalidValue;
    const long long blocks = (long long)((N + MG_BN - 1) / MG_BN) * (D / MG_BK);
    if (blocks > 0x7fffffffll) return hipErrorInvalidValue;
    hipLaunchKernelGGL(mocha_tile_bf16, dim3((unsigned)blocks), dim3(256), 0, s, (const unsigned short*)bank16, (unsigned short*)out, (long long)N, D);
    return hipGetLastError();
}

hipError_t launch_match_gemm_bf16(const void* qc16, const void* bank16, float* S, int Q, int64_t N, int D, int ksplit, hipStream_t s, int planes, const void* tiled,
                                  int nt_bank, unsigned* tickets) {
    if (Q <= 0 || N <= 0) return hipSuccess;
    if (D % MG_BK || (ksplit != 1 && ksplit != 2 && ksplit != 4 && ksplit != 8)) return hipErrorInvalidValue;
    if ((long long)MG_BM * D * 2 >= (1ll << 31)) return hipErrorInvalidValue;            // 32-bit buffer offsets inside a tile
    MatchGemmParams p;
    p.A = (const unsigned short*)qc16; p.B = (const unsigned short*)bank16; p.S = S;
    p.Q = Q; p.N = N; p.D = D; p.ksplit = ksplit; p.slab_stride = (long long)Q * N;
    p.m_tiles = (Q + MG_BM - 1) / MG_BM; p.n_tiles = (int)((N + MG_BN - 1) / MG_BN);
    p.a_plane = (long long)Q * D;
    p.Bt = (const unsigned short*)tiled;
    p.tickets = ksplit <= 8 ? tickets : nullptr;
    if (planes != 1 && planes != 2) return hipErrorInvalidValue;
    if (tiled && (long long)(D / MG_BK) * MG_BN * MG_BK * 2 >= (1ll << 31)) return hipErrorInvalidValue;      // 32-bit offsets inside a tile's block row
    const long long pairs = (long long)p.n_tiles * ksplit;
    const long long groups = (pairs + 7) / 8;
    // ring of 4 stages: three steps (96 KB) in flight per CU; measured equal to 3 and 5 stages, and 10 % faster than staging
    // through registers with ds_write (tools/experiments/README.md)
    // two query planes: stages of 48 KB, ring of 3 (two steps = 96 KB in flight, as before)
    constexpr size_t lds2 = mg_dma_lds_bytes<3, 2>(), lds1 = mg_dma_lds_bytes<4, 1>();
    const dim3 grid((unsigned)(groups * p.m_tiles * 8));
    if (planes == 2) {
        if (nt_bank) hipLaunchKernelGGL((mocha_match_gemm_bf16_dma<3, 2, true>), grid, dim3(256), lds2, s, p);
        else hipLaunchKernelGGL((mocha_match_gemm_bf16_dma<3, 2, false>), grid, dim3(256), lds2, s, p);
    } else {
        if (nt_bank) hipLaunchKernelGGL((mocha_match_gemm_bf16_dma<4, 1, true>), grid, dim3(256), lds1, s, p);
        else hipLaunchKernelGGL((mocha_match_gemm_bf16_dma<4, 1, false>), grid, dim3(256), lds1, s, p);
    }
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------------
// select: per query, every bank row whose coarse score lies within the coarse pass's ERROR BOUND of the best one is a
// candidate; the candidates' exact distances are evaluated in the direct form and the smallest wins (ties: lowest index).
//   coarse score  v_n = ||b_n - c||^2 - 2 S_n = d_n^2 - ||q - c||^2          (S summed over the K slices)
//   bound         |v_n(coarse) - v_n(exact)| <= rel (||q - c||^2 + ||b_n - c||^2):  bf16 query plane: |dq_i| <= 2^-9 |q_i|, so
//                 2 |sum dq_i b_i| <= 2^-8 ||q|| ||b|| <= 2^-9 (||q||^2 + ||b||^2) - rigorous, rel = 2^-9 plus slack for the fp32
//                 accumulation; exact-f32 MFMA pass: ~4e-7 (||q||^2 + ||b||^2) observed, rel = 4e-6.
//   candidates    v_n <= v_min + rel (2 ||q-c||^2 + ||b_n-c||^2 + ||b_min-c||^2)  -> the true nearest row is always among them.
//                 bf16 bank (round 3): the rounding part is priced with the query's MEASURED ||dq|| instead of its worst case:
//                 v_n <= v_min + 2 ||dq|| (||b_n-c|| + ||b_min-c||) + slack (...), ~0.4 of the margin above, about half the candidates.
//                 EVERY candidate is re-evaluated (no cap): the result is the exact search over the rows the kernel scans.
// One 1024-thread workgroup per query (a few hundred queries would not fill the chip with less).  The kernel is a chain of
// memory round trips, so every phase issues all of its loads before it uses any: score row (all K slices in flight) and
// the query row -> one block reduction (min score, ||q-c||^2) -> candidate list -> the candidates' bank rows, 16 waves
// shared among up to 16 candidates per pass, all of a wave's loads in flight at once.
// ---------------------------------------------------------------------------------------------------------------------
static constexpr int SEL_T = 1024, SEL_W = SEL_T / 64;
static constexpr int SEL_CH = 4;                    // score chunks (of 4 * SEL_T rows) held in registers
static constexpr int SEL_CAP = 128;                 // candidate list capacity per pass


__device__ __forceinline__ unsigned long long sel_key(float v, unsigned n) {          // order-preserving (value, index) key
    unsigned u = __float_as_uint(v);
    u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
    return ((unsigned long long)u << 32) | n;
}

__global__ __launch_bounds__(SEL_T) void mocha_match_select(const float* __restrict__ S, int ksplit, long long slab_stride, int lds,
                                                            const float* __restrict__ bnorm, const float* __restrict__ query,
                                                            const float* __restrict__ centre, const float* __restrict__ bank,
                                                            const unsigned short* __restrict__ bank16, long long N, int D,
                                                            float margin_rel, int32_t* __restrict__ idx, float* __restrict__ dist) {
    extern __shared__ __attribute__((aligned(16))) float sel_q[];      // [D]: q - c (bf16 bank) or q (fp32 bank)
    __shared__ unsigned long long rk[SEL_W];
    __shared__ float rs[SEL_W], rs2[SEL_W];
    __shared__ unsigned long long r_key;
    __shared__ float r_qn, r_dq;
    __shared__ int cand[SEL_CAP], csort[SEL_CAP];
    __shared__ float candv[SEL_CAP];
    __shared__ int ncand;
    __shared__ float dsum[SEL_W];
    __shared__ unsigned long long best;
    const int q = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) { ncand = 0; best = ~0ull; }

    // ---- 0. in flight while the scores are read: the query row goes straight into LDS (LDS-DMA, 1 KB per wave instruction,
    // no registers), the centroid into registers (at most 6 x 16 bytes per thread)
    const size_t qo = (size_t)q * D;
    constexpr int QN = 6;
#pragma unroll
    for (int i = 0; i < QN; ++i) {
        const int pi = i * SEL_W + wave;                         // 1 KB piece of the row
        if (pi * 256 < D)                                        // uniform per wave
            __builtin_amdgcn_global_load_lds(query + qo + pi * 256 + lane * 4, (__attribute__((address_space(3))) void*)(sel_q + pi * 256), 16, 0, 0);
    }
    f32x4 cv[QN];
#pragma unroll
    for (int i = 0; i < QN; ++i) {
        int e = (i * SEL_T + tid) * 4;
        e = e < D ? e : D - 4;
        cv[i] = *reinterpret_cast<const f32x4*>(centre + e);
    }

    // ---- 1. coarse scores; thread t owns rows c0 + 4 t .. 4 t + 3 of every chunk (one 16-byte load per K slice and chunk)
    const float* Sq = S + (size_t)q * lds;
    const bool vec = ((lds | (int)(N & 3)) & 3) == 0 && ((size_t)slab_stride & 3) == 0;
    auto scores = [&](long long c0, float (&v)[4], float (&bn)[4]) __attribute__((always_inline)) {
        const long long n0 = c0 + tid * 4;
        float dot[4] = {0.f, 0.f, 0.f, 0.f};
        if (vec) {
            const long long nb = n0 + 3 < N ? n0 : (N - 4 > 0 ? N - 4 : 0);          // clamped: loads are unconditional
#pragma unroll 8
            for (int z = 0; z < ksplit; ++z) {
                const f32x4 d = *reinterpret_cast<const f32x4*>(Sq + (size_t)z * slab_stride + nb);
                dot[0] += d[0]; dot[1] += d[1]; dot[2] += d[2]; dot[3] += d[3];
            }
            const f32x4 b = *reinterpret_cast<const f32x4*>(bnorm + nb);
            bn[0] = b[0]; bn[1] = b[1]; bn[2] = b[2]; bn[3] = b[3];
        } else {
            long long nc[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) nc[e] = n0 + e < N ? n0 + e : N - 1;
#pragma unroll 4
            for (int z = 0; z < ksplit; ++z)
#pragma unroll
                for (int e = 0; e < 4; ++e) dot[e] += Sq[(size_t)z * slab_stride + nc[e]];
#pragma unroll
            for (int e = 0; e < 4; ++e) bn[e] = bnorm[nc[e]];
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float sc = bn[e] - 2.f * dot[e];
            v[e] = (n0 + e < N && sc == sc) ? sc : INFINITY;      // rows past the end and NaN scores never qualify
        }
    };
    float v[SEL_CH][4], bn[SEL_CH][4];
    unsigned long long key = ~0ull;
#pragma unroll
    for (int ch = 0; ch < SEL_CH; ++ch) {
        const long long c0 = (long long)ch * SEL_T * 4;
        if (c0 < N) {                                            // uniform
            scores(c0, v[ch], bn[ch]);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const unsigned long long k = sel_key(v[ch][e], (unsigned)(c0 + tid * 4 + e));
                key = k < key ? k : key;
            }
        }
    }
    for (long long c0 = (long long)SEL_CH * SEL_T * 4; c0 < N; c0 += SEL_T * 4) {    // banks beyond 16 384 rows: not kept in registers
        float vv[4], bb[4];
        scores(c0, vv, bb);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const unsigned long long k = sel_key(vv[e], (unsigned)(c0 + tid * 4 + e));
            key = k < key ? k : key;
        }
    }
    // ---- 2. the centred query into LDS, ||q - c||^2, and the block reduction (best score, its row; the norm)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // this wave's pieces of the query row have landed ...
    __syncthreads();                                             // ... and everyone else's
    // bf16 bank: also ||dq||^2, dq = (q - c) - bf16(q - c) - exactly the rounding the coarse pass's query plane carries
    // (mocha_center_bf16 forms q - c with the same fp32 subtraction and the same round-to-nearest-even conversion)
    float qn = 0.f, dqn = 0.f;
#pragma unroll
    for (int i = 0; i < QN; ++i) {
        const int e = (i * SEL_T + tid) * 4;
        if (e < D) {
            const f32x4 qc = *reinterpret_cast<const f32x4*>(sel_q + e) - cv[i];
            qn = fmaf(qc[0], qc[0], qn); qn = fmaf(qc[1], qc[1], qn); qn = fmaf(qc[2], qc[2], qn); qn = fmaf(qc[3], qc[3], qn);
            if (bank16) {
                *reinterpret_cast<f32x4*>(sel_q + e) = qc;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float d = qc[k] - __uint_as_float(bf16_bits(qc[k]) << 16);
                    dqn = fmaf(d, d, dqn);
                }
            }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned long long k2 = __shfl_xor(key, o);
        key = k2 < key ? k2 : key;
        qn += __shfl_xor(qn, o);
        dqn += __shfl_xor(dqn, o);
    }
    if (lane == 0) { rk[wave] = key; rs[wave] = qn; rs2[wave] = dqn; }
    __syncthreads();
    if (wave == 0) {
        unsigned long long k = lane < SEL_W ? rk[lane] : ~0ull;
        float sq = lane < SEL_W ? rs[lane] : 0.f, sd = lane < SEL_W ? rs2[lane] : 0.f;
#pragma unroll
        for (int o = SEL_W / 2; o > 0; o >>= 1) {
            const unsigned long long k2 = __shfl_xor(k, o);
            k = k2 < k ? k2 : k;
            sq += __shfl_xor(sq, o);
            sd += __shfl_xor(sd, o);
        }
        if (lane == 0) { r_key = k; r_qn = sq; r_dq = sd; }
    }
    __syncthreads();
    const unsigned nmin = (unsigned)(r_key & 0xffffffffull);
    const unsigned umin = (unsigned)(r_key >> 32);
    const float vmin = __uint_as_float((umin & 0x80000000u) ? (umin & 0x7fffffffu) : ~umin);
    const bool none = !(vmin < INFINITY);                        // no finite score at all (NaN / inf inputs): row 0, distance NaN / inf
    const float bmin = none ? 0.f : bnorm[nmin];
    const float base = 2.f * r_qn + bmin;
    // bf16 bank: the coarse score of row n differs from the exact one by 2 |sum dq_i b_ni| <= 2 ||dq|| ||b_n - c|| (Cauchy-Schwarz with the
    // MEASURED ||dq|| of this query: about 0.4 of the worst case 2^-9 ||q - c|| the first version priced), plus margin_rel (...) for the
    // fp32 accumulation of the pass.  fp32 bank: margin_rel (...) alone, as before.
    const float dq2 = bank16 ? 2.000002f * sqrtf(r_dq) : 0.f;
    const float sbmin = sqrtf(bmin);

    // ---- 3. candidates; passes over index windows only when more than SEL_CAP rows qualify (each window then holds <= SEL_CAP)
    auto qualifies = [&](float sc, float b, unsigned n) -> bool {
        return n == nmin || sc <= vmin + dq2 * (sqrtf(b) + sbmin) + margin_rel * (base + b);
    };
    int total = 0;
    if (!none) {
#pragma unroll
        for (int ch = 0; ch < SEL_CH; ++ch) {
            const long long c0 = (long long)ch * SEL_T * 4;
            if (c0 < N) {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (v[ch][e] < INFINITY && qualifies(v[ch][e], bn[ch][e], (unsigned)(c0 + tid * 4 + e))) {
                        const int pos = atomicAdd(&ncand, 1);
                        if (pos < SEL_CAP) { cand[pos] = (int)(c0 + tid * 4 + e); candv[pos] = v[ch][e]; }
                    }
            }
        }
        for (long long c0 = (long long)SEL_CH * SEL_T * 4; c0 < N; c0 += SEL_T * 4) {
            float vv[4], bb[4];
            scores(c0, vv, bb);
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (vv[e] < INFINITY && qualifies(vv[e], bb[e], (unsigned)(c0 + tid * 4 + e))) {
                    const int pos = atomicAdd(&ncand, 1);
                    if (pos < SEL_CAP) { cand[pos] = (int)(c0 + tid * 4 + e); candv[pos] = vv[e]; }
                }
        }
    } else if (tid == 0) { cand[0] = 0; ncand = 1; }
    __syncthreads();
    total = ncand;
    const bool windowed = total > SEL_CAP;                       // uniform
    int wpc = SEL_W;                                             // waves per candidate row: 16 / pow2(min(candidates, 16))
    while (wpc > 1 && wpc * (total < SEL_W ? total : SEL_W) > SEL_W) wpc >>= 1;
    const long long nwin = windowed ? (N + SEL_CAP - 1) / SEL_CAP : 1;

    for (long long w = 0; w < nwin; ++w) {
        int nc = total;
        if (windowed) {                                          // rare: rebuild the list for rows [w CAP, (w + 1) CAP)
            __syncthreads();
            if (tid == 0) ncand = 0;
            __syncthreads();
            const long long lo = w * SEL_CAP, hi = lo + SEL_CAP;
            for (long long c0 = (lo / (SEL_T * 4)) * (SEL_T * 4); c0 < hi && c0 < N; c0 += SEL_T * 4) {
                float vv[4], bb[4];
                scores(c0, vv, bb);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const long long n = c0 + tid * 4 + e;
                    if (n >= lo && n < hi && vv[e] < INFINITY && qualifies(vv[e], bb[e], (unsigned)n)) {
                        const int pos = atomicAdd(&ncand, 1);
                        cand[pos] = (int)n; candv[pos] = vv[e];
                    }
                }
            }
            __syncthreads();
            nc = ncand;
            if (nc == 0) continue;                               // uniform
        }
        // rank sort by row index: the evaluation order (and with it every rounding) does not depend on the atomics' order
        if (tid < nc) {
            const int mine = cand[tid];
            int r = 0;
            for (int i = 0; i < nc; ++i) r += cand[i] < mine;
            csort[r] = mine;
        }
        __syncthreads();
        // ---- 4. exact squared distances: passes of up to 16 candidates.  The number of waves that share a row fixes the order
        // its terms are summed in, so it is chosen ONCE per query (from the total candidate count): every candidate of the query is
        // evaluated in the same order and identical rows get identical distances (ties then go to the lowest index).
        for (int p0 = 0; p0 < nc; p0 += SEL_W / wpc) {
            const int per = SEL_W / wpc;
            const int nb = nc - p0 < per ? nc - p0 : per;
            const int ci = wave / wpc, sub = wave % wpc;
            float a = 0.f;
            if (ci < nb) {                                       // uniform per wave
                const int row = csort[p0 + ci];
                const int lanes = wpc * 64, l = sub * 64 + lane;
                if (bank16) {
                    const u32x4* b = reinterpret_cast<const u32x4*>(bank16 + (size_t)row * D);
                    const int np = D / 8;                        // 16-byte pieces of 8 bf16
                    constexpr int NB = 12;
                    for (int it0 = 0; it0 * lanes < np; it0 += NB) {
                        u32x4 wv_[NB];
#pragma unroll
                        for (int u = 0; u < NB; ++u) {
                            int pc = (it0 + u) * lanes + l;
                            pc = pc < np ? pc : np - 1;
                            wv_[u] = __builtin_nontemporal_load(b + pc);
                        }
#pragma unroll
                        for (int u = 0; u < NB; ++u) {
                            const int pc = (it0 + u) * lanes + l;
                            if (pc < np) {
                                const f32x4 q0 = *reinterpret_cast<const f32x4*>(sel_q + pc * 8), q1 = *reinterpret_cast<const f32x4*>(sel_q + pc * 8 + 4);
                                const f32x4 b0 = {__uint_as_float(wv_[u][0] << 16), __uint_as_float(wv_[u][0] & 0xffff0000u), __uint_as_float(wv_[u][1] << 16), __uint_as_float(wv_[u][1] & 0xffff0000u)};
                                const f32x4 b1 = {__uint_as_float(wv_[u][2] << 16), __uint_as_float(wv_[u][2] & 0xffff0000u), __uint_as_float(wv_[u][3] << 16), __uint_as_float(wv_[u][3] & 0xffff0000u)};
                                const f32x4 d0 = q0 - b0, d1 = q1 - b1;
                                a = fmaf(d0[0], d0[0], a); a = fmaf(d0[1], d0[1], a); a = fmaf(d0[2], d0[2], a); a = fmaf(d0[3], d0[3], a);
                                a = fmaf(d1[0], d1[0], a); a = fmaf(d1[1], d1[1], a); a = fmaf(d1[2], d1[2], a); a = fmaf(d1[3], d1[3], a);
                            }
                        }
                    }
                } else {
                    const f32x4* b = reinterpret_cast<const f32x4*>(bank + (size_t)row * D);
                    const int np = D / 4;
                    constexpr int NB = 12;
                    for (int it0 = 0; it0 * lanes < np; it0 += NB) {
                        f32x4 wv_[NB];
#pragma unroll
                        for (int u = 0; u < NB; ++u) {
                            int pc = (it0 + u) * lanes + l;
                            pc = pc < np ? pc : np - 1;
                            wv_[u] = __builtin_nontemporal_load(b + pc);
                        }
#pragma unroll
                        for (int u = 0; u < NB; ++u) {
                            const int pc = (it0 + u) * lanes + l;
                            if (pc < np) {
                                const f32x4 d = *reinterpret_cast<const f32x4*>(sel_q + pc * 4) - wv_[u];
                                a = fmaf(d[0], d[0], a); a = fmaf(d[1], d[1], a); a = fmaf(d[2], d[2], a); a = fmaf(d[3], d[3], a);
                            }
                        }
                    }
                }
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o);
            }
            if (lane == 0) dsum[wave] = a;
            __syncthreads();
            if (tid == 0) {
                unsigned long long bk = best;
                for (int c = 0; c < nb; ++c) {
                    float d2 = 0.f;
                    for (int sw = 0; sw < wpc; ++sw) d2 += dsum[c * wpc + sw];
                    // distances are >= 0 (or NaN, which sorts last): the plain bit pattern orders them
                    const unsigned long long k = ((unsigned long long)__float_as_uint(d2) << 32) | (unsigned)csort[p0 + c];
                    bk = (c == 0 && bk == ~0ull) || k < bk ? k : bk;
                }
                best = bk;
            }
            __syncthreads();
        }
    }
    if (tid == 0) {
        idx[q] = (int)(best & 0xffffffffull);
        if (dist) dist[q] = sqrtf(__uint_as_float((unsigned)(best >> 32)));
    }
}

hipError_t match_select_init() {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(&mocha_match_select), hipFuncAttributeMaxDynamicSharedMemorySize, 23040 * 4);
}

hipError_t launch_match_select(const float* S, int ksplit, long long slab_stride, int lds, const float* bnorm, const float* query,
                               const float* centre, const float* bank, const void* bank16, float margin_rel, int Q, int64_t N, int D,
                               int32_t* idx, float* dist, hipStream_t s) {
    if (Q <= 0) return hipSuccess;
    if (D % 256 || D > 23040 || N < 1 || N > 0x7ffffff0ll || !(margin_rel >= 0.f)) return hipErrorInvalidValue;
    hipLaunchKernelGGL(mocha_match_select, dim3(Q), dim3(SEL_T), (size_t)D * sizeof(float), s, S, ksplit, slab_stride, lds, bnorm, query,
                       centre, bank, (const unsigned short*)bank16, (long long)N, D, margin_rel, idx, dist);
    return hipGetLastError();
}

}  // namespace mocha
