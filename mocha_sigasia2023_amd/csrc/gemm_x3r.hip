// The plane GEMM of gemm_x3.hip with the ACTIVATIONS RESIDENT IN REGISTERS (round 6): C = epilogue(A · W^T), K = 256, N a multiple of 128.
//
// mocha_gemm_x3 pays the activation path - fetch, split into three bf16 planes (44 VALU instructions per K step), three plane stores and
// their fragment reads - once per (m, n) tile: the 1 536-wide qkv projection fetches and splits the same 128 x 256 panel twelve times.
// Here a wave owns 32 rows of a 128-row panel for ALL the n-tiles of a unit: it fetches its rows once, splits them once and keeps the
// planes in the MFMA operand layout - 16 K steps x 3 planes x 4 registers = 192 of the 512 registers a lane has at one wave per SIMD.
// After that the K loop runs ACROSS n-tiles and moves only weights: the packed image of mocha_pack_x3 ([n tile][k step] blocks of 12 KB,
// consecutive for consecutive tiles) streams through a ring of four LDS stages by buffer_load ... lds, four steps ahead, one barrier per
// step; per step a wave reads 12 weight fragments (ds_read_b128, into the register set the NEXT step multiplies from) and issues
// 24 v_mfma_f32_32x32x16_bf16 - no VALU, no LDS store and no fragment read for the activations in the loop.
//
// Same plane values, same six products in the same order per K step, same K order per output element as mocha_gemm_x3: the accumulators
// are bit-identical (tests/test_gemm_engines.py); the epilogue (bias, GELU / LeakyReLU / ReLU) applies the same operations.
//
// Work units: panels are handed out whole (all n-tiles) while whole rounds of the grid last; the remaining panels are cut into chunks of
// `tail_chunk` n-tiles so that the last round is not a quarter-full chip (823 panels over 256 workgroups = 3.2 rounds).
#include "kernels.h"
#include "device_utils.h"
#include <algorithm>

namespace mocha {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

#ifndef X3R_STORE_AUX
#define X3R_STORE_AUX 0         // cache policy bits of the output stores (2 = nt)
#endif
static constexpr int RK = 256, RSTEPS = RK / 16;            // contraction length held in registers: 16 K steps
static constexpr int RB_HALF = 128 * 8 + 32;                // bf16 per k half of a weight plane in LDS (the padded halves of gemm_x3.hip)
static constexpr int RB_PLANE = 2 * RB_HALF;
static constexpr int R_STAGE = 3 * RB_PLANE;                // 6 528 bf16 = 13 056 B
#ifndef X3R_RING
#define X3R_RING 4
#endif
static constexpr int R_RING = X3R_RING;                     // LDS stages (a power of two); the copy of step g + R_RING is issued during step g
static constexpr int R_WAIT = (R_RING - 2) * 3;             // copies that may stay in flight at the end of a step: those of steps g + 3 .. g + R_RING
static_assert((R_RING & (R_RING - 1)) == 0 && R_RING >= 4 && R_RING <= 8 && R_WAIT + 16 < 64, "ring");
static constexpr int RW_BLOCK = 3 * 128 * 16;               // packed weights per (n tile, k step): 6 144 bf16 = 12 KB (XW_BLOCK)
static constexpr int R_LDS_BYTES = R_RING * R_STAGE * 2;    // 52 224 B

// diagnostic build (tools/): -DX3R_STAMPS accumulates, per wave, the shader cycles spent (0 -> 1) in the end-of-step counted wait, (1 -> 2) in
// the barrier and (3 -> 0) issuing a step's MFMAs / reads / copies, plus the unit prologue and the tile epilogues, into the buffer at p.wsub
#ifdef X3R_STAMPS
#define X3R_T(i) do { const long long now__ = (long long)__builtin_readcyclecounter(); st_acc[i] += now__ - st_last; st_last = now__; } while (0)
#else
#define X3R_T(i)
#endif

struct X3rUnits { int whole; int tail_chunk; int total; };  // panels handed out whole; n-tiles per tail unit; number of units

// EPI: the epilogue adds a bias and applies p.act (the bias quads of a tile are fetched during its ninth K step; the counted waits count them)
template <bool EPI>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void mocha_gemm_x3r(GemmParams p, X3rUnits un) {
    extern __shared__ __attribute__((aligned(16))) unsigned short xr_sm[];          // [R_RING][R_STAGE]
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // wave-uniform: the copies' LDS addresses stay scalar
    const int l31 = lane & 31, hh = lane >> 5;
    const int n_tiles = p.N / 128;
    const int chunks = (n_tiles + un.tail_chunk - 1) / un.tail_chunk;

    // fragment of lane (column l31 of a 32-column block, k half hh): 16 bytes of a plane
    const int fb = hh * RB_HALF + l31 * 8;

#ifdef X3R_STAMPS
    long long st_acc[6] = {0, 0, 0, 0, 0, 0}, st_last = (long long)__builtin_readcyclecounter();
    const long long st_begin = st_last, rt_begin = (long long)__builtin_amdgcn_s_memrealtime();
#endif
    for (int u = blockIdx.x; u < un.total; u += gridDim.x) {
        int panel, t0, nt;
        if (u < un.whole) { panel = u; t0 = 0; nt = n_tiles; }
        else {
            const int v = u - un.whole;
            panel = un.whole + v / chunks;
            t0 = (v - (v / chunks) * chunks) * un.tail_chunk;
            nt = n_tiles - t0 < un.tail_chunk ? n_tiles - t0 : un.tail_chunk;
        }
        const int m0 = panel * 128;
        const int total_steps = nt * RSTEPS;
        const __amdgpu_buffer_rsrc_t rsW = make_rsrc(p.Wsplit + (size_t)t0 * RSTEPS * RW_BLOCK);
        // weights of step g (0 .. total_steps - 1 of this unit) into ring stage g & 3: piece j * 4 + wave of the packed block = (plane, k half, 64-row half)
        auto dma_piece = [&](int g, int j) __attribute__((always_inline)) {
            const int gg = g < total_steps ? g : total_steps - 1;           // past the end: a block nobody reads, into a free stage (the counted waits stay the same)
            unsigned short* st = xr_sm + (g & (R_RING - 1)) * R_STAGE;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (__attribute__((address_space(3))) void*)(st + ((j * 4 + wave) >> 2) * RB_PLANE +
                                                     (((j * 4 + wave) >> 1) & 1) * RB_HALF + ((j * 4 + wave) & 1) * 512), 16,
                                                     (unsigned)(j * 256 + tid) * 16u, (unsigned)gg * (RW_BLOCK * 2u), 0, 0);
        };
        // the first four weight blocks land while the activations are fetched and split
#pragma unroll
        for (int g = 0; g < R_RING; ++g)
#pragma unroll
            for (int j = 0; j < 3; ++j) dma_piece(g, j);

        // ---- this wave's 32 rows, once: lane (row l31, k half hh) fetches the 32 bytes of every K step and keeps them as three planes
        s16x8 ap[RSTEPS][3];
        {
            int row = m0 + wave * 32 + l31;
            row = row < p.M ? row : p.M - 1;
            const __amdgpu_buffer_rsrc_t rsA = make_rsrc(p.A + (size_t)m0 * p.lda);
            const unsigned off = (unsigned)(row - m0) * (unsigned)p.lda * 4u + (unsigned)hh * 32u;
            f32x4 raw[2][8];
            auto fetch = [&](int b, f32x4 (&r)[8]) __attribute__((always_inline)) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    r[2 * k] = bload(rsA, off, (unsigned)(b * 4 + k) * 64u);
                    r[2 * k + 1] = bload(rsA, off + 16u, (unsigned)(b * 4 + k) * 64u);
                }
            };
            fetch(0, raw[0]);
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                if (b + 1 < 4) fetch(b + 1, raw[(b + 1) & 1]);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const f32x4 lo = raw[b & 1][2 * k], hi = raw[b & 1][2 * k + 1];
                    float x[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                    if (p.a_lrelu) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) x[e] = x[e] > 0.f ? x[e] : 0.2f * x[e];
                    }
                    plane_split8(x, ap[b * 4 + k]);
                }
            }
        }

        // ---- the first step's fragments
        s16x8 bf[2][3][4];
        asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"((R_RING - 1) * 3) : "memory");      // block 0 has landed for every wave (the others stay in flight)
#pragma unroll
        for (int q = 0; q < 3; ++q)
#pragma unroll
            for (int j = 0; j < 4; ++j) bf[0][q][j] = *reinterpret_cast<const s16x8*>(xr_sm + q * RB_PLANE + fb + j * 32 * 8);
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(R_WAIT) : "memory");  // block 1 has landed; stage 0 is free again
        // (the copy of step R_RING goes into stage 0 during step 0, below)

        X3R_T(4);                                            // unit prologue
        const __amdgpu_buffer_rsrc_t rsBias = make_rsrc(p.bias ? p.bias : p.A);
        const int rows_valid = p.M - m0 < 128 ? p.M - m0 : 128;
        f32x16 acc[4];
        f32x4 bq[16];
        for (int t = 0; t < nt; ++t) {
            const int gbase = t * RSTEPS;
#pragma unroll
            for (int s = 0; s < RSTEPS; ++s) {
                const int P = s & 1;
                const unsigned short* nxt = xr_sm + ((s + 1) & (R_RING - 1)) * R_STAGE;
                X3R_T(3);
#pragma unroll
                for (int m = 0; m < 24; ++m) {
                    const int pr = m >> 2, j = m & 3, pa = PLANE_PA[pr], pb = PLANE_PB[pr];
                    if (s == 0 && pr == 0) {
                        const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf[P][pb][j], ap[s][pa], z, 0, 0, 0);
                    } else
                        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf[P][pb][j], ap[s][pa], acc[j], 0, 0, 0);
                    // the next step's fragments, one read per MFMA; then the copy of the step four ahead into the stage this step's fragments came from
                    if (m < 12) bf[P ^ 1][m >> 2][m & 3] = *reinterpret_cast<const s16x8*>(nxt + (m >> 2) * RB_PLANE + fb + (m & 3) * 32 * 8);
                    else if (m < 15) dma_piece(gbase + s + R_RING, m - 12);
                    else if (EPI && s == 8 && m < 19) {             // this tile's bias quads: four per MFMA, after the step's copies
#pragma unroll
                        for (int g = 0; g < 4; ++g) bq[(m - 15) * 4 + g] = bload(rsBias, (unsigned)((t0 + t) * 128 + (m - 15) * 32 + 8 * g + 4 * hh) * 4u, 0u);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                // step s + 2's weights have landed (s + 3, s + 4 in flight) and this wave's fragment reads are done; the first two steps after a tile's for the same copies, which are OLDER than the 16 stores (the counter retires in order)
                // stores wait for the same copies, which are OLDER than the 16 stores (the counter retires in order); likewise the 16 bias fetches
                X3R_T(0);
                if ((t > 0 && s < R_RING - 2) || (EPI && s >= 8 && s <= 8 + R_RING - 2)) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(R_WAIT + 16) : "memory");
                else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(R_WAIT) : "memory");
                X3R_T(1);
#ifndef X3R_NOBARRIER
                asm volatile("s_barrier" ::: "memory");
#endif
                X3R_T(2);
            }
            // ---- epilogue of tile t0 + t: straight from the accumulators (lane = row l31 of the wave's block, registers 4g .. 4g+3 = columns 8g + 4hh + e)
            const int n0 = (t0 + t) * 128;
            // rows past M: the buffer's size drops them (the stores stay unconditional: the counted waits above count them)
            const __amdgpu_buffer_rsrc_t rsC = __builtin_amdgcn_make_buffer_rsrc(p.C + (size_t)m0 * p.ldc, 0, (int)((unsigned)rows_valid * (unsigned)p.ldc * 4u), 0x00020000);
            const unsigned crow = (unsigned)(wave * 32 + l31) * (unsigned)p.ldc * 4u + (unsigned)(n0 + 4 * hh) * 4u;
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    f32x4 v = {acc[j][4 * g], acc[j][4 * g + 1], acc[j][4 * g + 2], acc[j][4 * g + 3]};
                    if (EPI) {
                        v += bq[j * 4 + g];
                    if (p.act == 1) v = mocha_gelu4(v);
                    else if (p.act == 2) { v[0] = v[0] > 0.f ? v[0] : 0.2f * v[0]; v[1] = v[1] > 0.f ? v[1] : 0.2f * v[1]; v[2] = v[2] > 0.f ? v[2] : 0.2f * v[2]; v[3] = v[3] > 0.f ? v[3] : 0.2f * v[3]; }
                    else if (p.act == 3) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f); }
                    }
                    bstore_aux<X3R_STORE_AUX>(rsC, v, crow + (unsigned)(j * 32 + 8 * g) * 4u, 0u);
                }
            X3R_T(5);                                        // tile epilogue
        }
        // the unit's last copies (clamped repeats) and stores: drained before the ring is reused
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        X3R_T(4);
    }
#ifdef X3R_STAMPS
    if (lane == 0 && p.wsub) {
        long long* d = (long long*)p.wsub + ((size_t)blockIdx.x * 4 + wave) * 8;
        for (int i = 0; i < 6; ++i) d[i] = st_acc[i];
        d[6] = (long long)__builtin_readcyclecounter() - st_begin;
        d[7] = (long long)__builtin_amdgcn_s_memrealtime() - rt_begin;
    }
#endif
}

bool gemm_x3r_supports(const GemmParams& p) {
    if (p.K != RK || p.N % 128 != 0 || p.N < 256) return false;
#ifndef X3R_STAMPS
    if (p.wsub) return false;
#endif
    if (p.gather || p.ksplit > 1 || p.residual || p.rowbias) return false;
    if (p.act && !p.bias) return false;           // the epilogue instance is the one with a bias
    if ((p.ldc & 3) || (p.lda & 3)) return false;
    if (128ll * p.lda * 4 >= (1ll << 31) || 128ll * p.ldc * 4 >= (1ll << 31)) return false;
    return p.M >= 128 * 64;                       // a chip's worth of panels; smaller launches keep the tiled instances
}

hipError_t gemm_x3r_init() {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&mocha_gemm_x3r<false>), hipFuncAttributeMaxDynamicSharedMemorySize, R_LDS_BYTES);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&mocha_gemm_x3r<true>), hipFuncAttributeMaxDynamicSharedMemorySize, R_LDS_BYTES);
    return e;
}

// units: whole panels while whole rounds of the grid last, the rest in chunks of the n-tile count that minimises (rounds x (chunk + prologue))
hipError_t launch_gemm_x3r(const GemmParams& p, hipStream_t s, int grid) {
    if (!p.Wsplit || !gemm_x3r_supports(p)) return hipErrorInvalidValue;
    if (grid <= 0) grid = 256;
    const int panels = (p.M + 127) / 128, n_tiles = p.N / 128;
    X3rUnits un;
    un.whole = panels / grid * grid;
    const int rest = panels - un.whole;
    un.tail_chunk = n_tiles;
    double best = 1e30;
    for (int c = 1; c <= n_tiles; ++c) {
        const int units = rest * ((n_tiles + c - 1) / c);
        const double cost = (double)((units + grid - 1) / grid) * (c + 0.5);      // a unit's prologue: about half a tile's K loop
        if (cost < best) { best = cost; un.tail_chunk = c; }
    }
    un.total = un.whole + rest * ((n_tiles + un.tail_chunk - 1) / un.tail_chunk);
    if (p.bias) hipLaunchKernelGGL(mocha_gemm_x3r<true>, dim3((unsigned)std::min(grid, un.total)), dim3(256), R_LDS_BYTES, s, p, un);
    else hipLaunchKernelGGL(mocha_gemm_x3r<false>, dim3((unsigned)std::min(grid, un.total)), dim3(256), R_LDS_BYTES, s, p, un);
    return hipGetLastError();
}

}  // namespace mocha
