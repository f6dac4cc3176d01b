// The plane GEMM of gemm_x3.hip with the ACTIVATIONS RESIDENT IN REGISTERS (round 6): C = epilogue(A · W^T), K = 256, N a multiple of 128.
//
// mocha_gemm_x3 pays the activation path - fetch, split into three bf16 planes (44 VALU instructions per K step), three plane stores and
// their fragment reads - once per (m, n) tile: the 1 536-wide qkv projection fetches and splits the same 128 x 256 panel twelve times.
// Here a wave owns 32 rows of a 128-row panel for ALL the n-tiles of a unit: it fetches its rows once, splits them once and keeps the
// planes in the MFMA operand layout - 16 K steps x 3 planes x 4 registers = 192 of the 512 registers a lane has at one wave per SIMD.
// After that the K loop runs ACROSS n-tiles (and across units) and moves only weights: the packed image of mocha_pack_x3 ([n tile][k step]
// blocks of 12 KB) streams through a ring of LDS stages by buffer_load ... lds, a ring's length ahead, one barrier per step; per step a
// wave reads 12 weight fragments (ds_read_b128, into the register set the NEXT step multiplies from) and issues 24
// v_mfma_f32_32x32x16_bf16 - no VALU, no LDS store and no fragment read for the activations in the loop.
//
// One wave per SIMD has nobody to hide behind, so nothing in the loop may wait (v1 of this file, profiles/r06/b_x3r_v1_variants_stamps.txt:
// 20 % of the time in tile epilogues - every CU storing its tile at the same moment - and 8 % in unit prologues):
//   * two accumulator sets: tile t's 16 stores per lane are issued one per K step of tile t + 1 (the output leaves the chip evenly, not in
//     bursts), its bias + activation arithmetic spread between that step's MFMAs;
//   * the weight ring and the fragment double buffer run on through tile and unit boundaries (the next unit's first blocks are copied
//     during the last steps of the current one): a unit change costs the fetch + split of the new rows and nothing else;
//   * every vector-memory operation in the loop is either a copy into LDS or a store, so the end-of-step wait is a COUNTED wait for the
//     copy of step g + 2 (the bias lives in LDS).
//
// Same plane values, same six products in the same order per K step, same K order per output element as mocha_gemm_x3: the accumulators
// are bit-identical (tools/gemm_bench mode 37 compares every element); the epilogue (bias, GELU / LeakyReLU / ReLU) applies the same operations.
//
// Work: the launch's tile PAIRS in (panel, tile) order are cut into gridDim.x contiguous ranges that differ by at most one pair, so every
// workgroup finishes at the same time (823 panels over 256 workgroups would otherwise be 3.2 rounds); a range's pieces of panels are its
// units - a first partial panel, whole panels, a last partial one.  The weight stream does not depend on the panel: consecutive tiles are
// consecutive blocks of the image, wrapping to block 0 at a panel's end.
#include "kernels.h"
#include "device_utils.h"
#include <algorithm>
#include <type_traits>

namespace mocha {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

#ifndef X3R_STORE_AUX
#define X3R_STORE_AUX 0         // cache policy bits of the output stores (2 = nt: measured 25 % slower, profiles/r06/b_x3r_v1_variants_stamps.txt)
#endif
static constexpr int RK = 256, RSTEPS = RK / 16;            // contraction length held in registers: 16 K steps
static constexpr int RB_HALF = 128 * 8 + 32;                // bf16 per k half of a weight plane in LDS (the padded halves of gemm_x3.hip)
static constexpr int RB_PLANE = 2 * RB_HALF;
static constexpr int R_STAGE = 3 * RB_PLANE;                // 6 528 bf16 = 13 056 B
#ifndef X3R_RING
#define X3R_RING 8
#endif
static constexpr int R_RING = X3R_RING;                     // LDS stages (a power of two dividing 16)
// The copy of step g + R_LEAD is issued during step g, into the stage step g - 1 multiplied from.  Step g + 1's fragments are read DURING step g
// (each plane's registers as soon as its last product of step g has been issued) and consumed by step g + 1's MFMAs before its barrier, and
// a stage is overwritten only by a copy issued two barriers after the step that read it: no wait for LDS reads is needed before a barrier.
static constexpr int R_LEAD = R_RING - 1;
// vector-memory operations that may stay in flight at the end of step g, when the copy of step g + 2 must have landed: the copies of steps
// g + 3 .. g + R_LEAD (three each) and the one store of each of the steps g + 3 - R_LEAD .. g (issued after the copy of step g + 2 was)
static constexpr int R_WAIT = (R_LEAD - 2) * 3 + (R_LEAD - 1);
static_assert((R_RING & (R_RING - 1)) == 0 && R_RING >= 4 && R_RING <= 8 && R_WAIT < 64, "ring");
static constexpr int RW_BLOCK = 3 * 128 * 16;               // packed weights per (n tile, k step): 6 144 bf16 = 12 KB (XW_BLOCK)
static constexpr int R_BIAS_MAX = 2048;                     // widest launch of the epilogue instance: its bias vector lives in LDS
static constexpr int R_LDS_BYTES = R_RING * R_STAGE * 2 + R_BIAS_MAX * 4 + 1024;      // + 256 B per wave where the prefetch copies land (never read)

// diagnostic build (tools/): -DX3R_STAMPS accumulates, per wave, the shader cycles (0) issuing a step's MFMAs / reads / copies / store,
// (1) in the end-of-step counted wait, (2) in the barrier, (4) in the unit changes (fetch + split) into the buffer at p.wsub
// (X3R_T(i) closes an interval and books it under i)
#ifdef X3R_STAMPS
#define X3R_T(i) do { const long long now__ = (long long)__builtin_readcyclecounter(); st_acc[i] += now__ - st_last; st_last = now__; } while (0)
#else
#define X3R_T(i)
#endif

// The matrix instruction with the register classes fixed by hand: accumulators and weight fragments in VGPRs, the resident activation
// planes in AGPRs (as an MFMA source operand an AGPR costs nothing on gfx950).  Left to the register allocator the kernel's 370 live
// registers - 192 of planes, 128 of accumulators, 48 of fragments - end up shuffled between the two halves of the file with copies in the
// K steps and spills to scratch (the compiler allocates MFMA sources in VGPRs only and uses AGPRs as spill slots).
// Hazards the compiler would otherwise pad (it does not look inside the asm): an accumulator is re-read as SrcC only by its own next
// product, three other MFMAs later (in-place accumulation, interlocked), and by a store a whole K step or more after its last product.
#define X3R_MFMA(acc, b, a) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(b), "a"(a))
#define X3R_MFMA0(acc, b, a) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(acc) : "v"(b), "a"(a))

// mocha_gelu4 (device_utils.h) cut into eight pieces of about six instructions, one per MFMA gap: the same operations on the same values,
// hence the same bits.  st: 0 |x| c and the polynomial's first steps ... 7 the last products.
struct X3rGelu4 {
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 x[2], t[2], q[2];
    float h[4];
    template <int ST> __device__ __forceinline__ void stage() {
        constexpr float C[10] = {1.146809295e-05f, -1.515590512e-04f, 8.423155240e-04f, -2.261537520e-03f, 6.770915453e-05f, 2.773738608e-02f,
                                 -1.483134404e-01f, -9.184416673e-01f, -1.627907386e+00f, -9.999999969e-01f};
        auto fma_from = [&](int a, int b) __attribute__((always_inline)) {                // polynomial steps a .. b - 1 (coefficients C[a] .. ) on both pairs
#pragma unroll
            for (int k = a; k < b; ++k)
#pragma unroll
                for (int hp = 0; hp < 2; ++hp) q[hp] = __builtin_elementwise_fma(q[hp], t[hp], (f2){C[k], C[k]});
        };
        if constexpr (ST == 0) {
#pragma unroll
            for (int hp = 0; hp < 2; ++hp) { t[hp] = __builtin_elementwise_abs(x[hp]) * 0.70710678118654752440f; q[hp] = (f2){C[0], C[0]}; }
            fma_from(1, 3);
        } else if constexpr (ST == 1) fma_from(3, 6);
        else if constexpr (ST == 2) fma_from(6, 9);
        else if constexpr (ST == 3) {
            fma_from(9, 10);
#pragma unroll
            for (int e = 0; e < 4; ++e) h[e] = __builtin_amdgcn_exp2f(q[e >> 1][e & 1]);
        } else if constexpr (ST == 4) {
#pragma unroll
            for (int e = 0; e < 4; ++e) h[e] = t[e >> 1][e & 1] > 4.3f ? 0.f : h[e];
        } else if constexpr (ST == 5) {
#pragma unroll
            for (int e = 0; e < 2; ++e) h[e] = x[0][e] * (x[0][e] < 0.f ? h[e] : 1.0f - h[e]);
        } else if constexpr (ST == 6) {
#pragma unroll
            for (int e = 2; e < 4; ++e) h[e] = x[1][e & 1] * (x[1][e & 1] < 0.f ? h[e] : 1.0f - h[e]);
        }
    }
};

struct X3rUnits { int ppp; long long pairs; int whole; };     // tile pairs per panel (N / 256); tile pairs of the launch; whole panels every workgroup takes first

// ACT: -1 no epilogue arithmetic; 0 the epilogue adds a bias; 1 bias + exact-erf GELU (compile-time: the quad's arithmetic sits in fixed MFMA gaps)
template <int ACT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void mocha_gemm_x3r(GemmParams p, X3rUnits un) {
    extern __shared__ __attribute__((aligned(16))) unsigned short xr_sm[];          // [R_RING][R_STAGE] bf16, then R_BIAS_MAX floats
    constexpr bool EPI = ACT >= 0;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // wave-uniform: the copies' LDS addresses stay scalar
    const int l31 = lane & 31, hh = lane >> 5;
    const int n_tiles = p.N / 128;
    float* bias_sm = reinterpret_cast<float*>(xr_sm + R_RING * R_STAGE);
#ifdef X3R_STAMPS
    long long st_acc[6] = {0, 0, 0, 0, 0, 0}, st_last = (long long)__builtin_readcyclecounter();
    const long long st_begin = st_last, rt_begin = (long long)__builtin_amdgcn_s_memrealtime();
#endif
    // this workgroup's tile pairs: first `whole` panels of its own (every workgroup starts them at tile 0 at the same time: one stream of
    // weight blocks for the chip), then its share of the remaining panels' pairs (ranges that differ by at most one pair).  The unit at
    // the front of the current range: (panel, first tile, tiles).
    const long long tail0 = (long long)gridDim.x * un.whole * un.ppp, tail_n = un.pairs - tail0;
    const long long seg1_lo = tail0 + (long long)blockIdx.x * tail_n / gridDim.x, seg1_hi = tail0 + (long long)(blockIdx.x + 1) * tail_n / gridDim.x;
    long long pr_cur = (long long)blockIdx.x * un.whole * un.ppp, pr_end = pr_cur + (long long)un.whole * un.ppp;
    bool in_tail = false;
    if (pr_cur >= pr_end) { pr_cur = seg1_lo; pr_end = seg1_hi; in_tail = true; }
    if (pr_cur >= pr_end) return;
    auto front_unit = [&](int& panel, int& t0, int& nt) __attribute__((always_inline)) {
        panel = (int)(pr_cur / un.ppp);
        const int tp = (int)(pr_cur - (long long)panel * un.ppp);
        const long long left = pr_end - pr_cur;
        const int np = un.ppp - tp < left ? un.ppp - tp : (int)left;
        t0 = 2 * tp; nt = 2 * np;
    };

    // ---- the weight stream of this workgroup: the blocks of its tiles one after the other, wrapping to block 0 after a panel's last tile.
    // Cursor = the next block to copy (all scalar).  Past the last tile the cursor repeats a block into stages nobody reads: the counted
    // waits stay the same.
    const __amdgpu_buffer_rsrc_t rsW = make_rsrc(p.Wsplit);                      // the whole image: N x 256 x 6 B <= 3 MB
    const int nblk = n_tiles * RSTEPS;
    int d_blk = (int)(pr_cur % un.ppp) * 2 * RSTEPS;
    long long d_left = (pr_end - pr_cur) * 2 * RSTEPS;
    long long d_next = in_tail ? 0 : (seg1_hi - seg1_lo) * 2 * RSTEPS;            // blocks of the range after the current one
    auto dma_advance = [&]() __attribute__((always_inline)) {                     // after a block's three pieces
        --d_left;
        const int nb = d_blk + 1 == nblk ? 0 : d_blk + 1;
        if (d_left > 0) d_blk = nb;
        else if (d_next > 0) { d_blk = (int)(seg1_lo % un.ppp) * 2 * RSTEPS; d_left = d_next; d_next = 0; }
    };
    // piece j * 4 + wave of the packed block = (plane, k half, 64-row half), into ring stage `stage`
    auto dma_piece = [&](int stage, int j) __attribute__((always_inline)) {
        unsigned short* st = xr_sm + stage * R_STAGE;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (__attribute__((address_space(3))) void*)(st + ((j * 4 + wave) >> 2) * RB_PLANE +
                                                 (((j * 4 + wave) >> 1) & 1) * RB_HALF + ((j * 4 + wave) & 1) * 512), 16,
                                                 (unsigned)(j * 256 + tid) * 16u, (unsigned)d_blk * (RW_BLOCK * 2u), 0, 0);
    };

    // ---- a wave's 32 rows of a panel, once per unit: lane (row l31, k half hh) fetches the 32 bytes of every K step and keeps them as three planes
    s16x8 ap[RSTEPS][3];
    auto load_a = [&](int panel) __attribute__((always_inline)) {
        const int m0 = panel * 128;
        int row = m0 + wave * 32 + l31;
        row = row < p.M ? row : p.M - 1;
        const __amdgpu_buffer_rsrc_t rsA = make_rsrc(p.A + (size_t)m0 * p.lda);
        const unsigned off = (unsigned)(row - m0) * (unsigned)p.lda * 4u + (unsigned)hh * 32u;
        f32x4 raw[3][8];                 // three of the four batches of eight fetches in flight (a unit change waits for memory about once, not four times)
        auto fetch = [&](int b, f32x4 (&r)[8]) __attribute__((always_inline)) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                r[2 * k] = bload(rsA, off, (unsigned)(b * 4 + k) * 64u);
                r[2 * k + 1] = bload(rsA, off + 16u, (unsigned)(b * 4 + k) * 64u);
            }
        };
        fetch(0, raw[0]); fetch(1, raw[1]); fetch(2, raw[2]);
#pragma unroll
        for (int b = 0; b < 4; ++b) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const f32x4 lo = raw[b % 3][2 * k], hi = raw[b % 3][2 * k + 1];
                float x[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                if (p.a_lrelu) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) x[e] = x[e] > 0.f ? x[e] : 0.2f * x[e];
                }
                s16x8 pl[3];
                plane_split8(x, pl);
#pragma unroll
                for (int q = 0; q < 3; ++q) asm volatile("; plane -> agpr" : "=a"(ap[b * 4 + k][q]) : "0"(pl[q]));      // from here on an AGPR tuple
            }
            if (b == 0) fetch(3, raw[0]);
        }
        asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");        // (the copies into AGPRs are VALU writes; the MFMAs that read them are opaque to the hazard padding)
    };

    int panel, t0, nt;
    front_unit(panel, t0, nt);
    // the first R_LEAD weight blocks land while the first unit's activations are fetched and split
#pragma unroll
    for (int g = 0; g < R_LEAD; ++g) {
#pragma unroll
        for (int j = 0; j < 3; ++j) dma_piece(g, j);
        dma_advance();
    }
    if (EPI) {
        for (int i = tid; i < p.N; i += 256) bias_sm[i] = p.bias[i];
    }
    load_a(panel);

    // fragment of lane (column l31 of a 32-column block, k half hh): 16 bytes of a plane
    const int fb = hh * RB_HALF + l31 * 8;
    s16x8 bf[3][4];
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");       // (the fetch of the activations has drained the copies anyway)
#pragma unroll
    for (int q = 0; q < 3; ++q)
#pragma unroll
        for (int j = 0; j < 4; ++j) bf[q][j] = *reinterpret_cast<const s16x8*>(xr_sm + q * RB_PLANE + fb + j * 32 * 8);

    // ---- the tile whose stores are pending: issued one per K step of the tile after it
    const float* pend_base = p.C;       // its panel's first row
    int pend_rows = 0;                  // rows of that panel inside M; 0 = nothing pending (the store is issued all the same and dropped by the buffer's size)
    int pend_n0 = 0;
    const unsigned crow0 = (unsigned)(wave * 32 + l31) * (unsigned)p.ldc * 4u + (unsigned)(4 * hh) * 4u;
    f32x16 acc[2][4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[1][j][r] = 0.f;          // the first tile's "pending" set: stored nowhere (pend_rows = 0), but defined

    // epilogue arithmetic of one quad (the same operations as mocha_gemm_x3's epilogue)
    auto finish = [&](f32x4 v, const f32x4 b4) __attribute__((always_inline)) -> f32x4 {
        if (ACT >= 0) v += b4;
        if (ACT == 1) v = mocha_gelu4(v);
        return v;
    };

    // ---- the next unit's rows are pulled into L2 while the current unit's last tile is multiplied: four copies of one dword per lane
    // (a lane per 128-byte line of the wave's 32 rows) into an LDS scratch nobody reads.  They are ordinary entries of the vector-memory
    // counter, NEWER than the copy the end-of-step wait is for: the counted wait stays sufficient (it only waits a little longer).
    const float* pf_base = p.A;
    int pf_rows = 0;                    // rows of the next panel inside M; 0 = nothing to prefetch
    auto prefetch = [&](int k) __attribute__((always_inline)) {
        const __amdgpu_buffer_rsrc_t rsN = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(pf_base), 0, (int)((unsigned)pf_rows * (unsigned)p.lda * 4u), 0x00020000);
        const int line = k * 64 + lane;                                       // 0 .. 255: (row, 128-byte line) = (line >> 3, line & 7)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsN, (__attribute__((address_space(3))) void*)(reinterpret_cast<char*>(bias_sm + R_BIAS_MAX) + wave * 256), 4,
                                                 (unsigned)(wave * 32 + (line >> 3)) * (unsigned)p.lda * 4u + (unsigned)(line & 7) * 128u, 0u, 0, 0);
    };

    // One tile: 16 K steps into acc[PAR] while acc[PAR ^ 1] - the tile before - leaves, one quad per step.
    auto tile = [&](auto par) __attribute__((always_inline)) {
        constexpr int PAR = decltype(par)::value;
        const __amdgpu_buffer_rsrc_t rsP = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(pend_base), 0, (int)((unsigned)pend_rows * (unsigned)p.ldc * 4u), 0x00020000);
#pragma unroll
        for (int s = 0; s < RSTEPS; ++s) {
            const unsigned short* nxt = xr_sm + ((s + 1) & (R_RING - 1)) * R_STAGE;
            const int ej = s >> 2, eg = s & 3;                      // the pending tile's quad of this step: columns ej * 32 + 8 eg + 4 hh ..
            f32x4 b4 = {0.f, 0.f, 0.f, 0.f};
            X3rGelu4 ge;
            f32x4 o4 = {acc[PAR ^ 1][ej][4 * eg], acc[PAR ^ 1][ej][4 * eg + 1], acc[PAR ^ 1][ej][4 * eg + 2], acc[PAR ^ 1][ej][4 * eg + 3]};
            X3R_T(3);
#pragma unroll
            for (int m = 0; m < 24; ++m) {
                const int pr = m >> 2, j = m & 3, pa = PLANE_PA[pr], pb = PLANE_PB[pr];
                if (EPI && m == 0) b4 = *reinterpret_cast<const f32x4*>(bias_sm + pend_n0 + ej * 32 + 8 * eg + 4 * hh);      // consumed twelve MFMAs later
                if (EPI && m == 12) o4 += b4;
                if (ACT == 1) {                                     // the quad's GELU, a few instructions per MFMA gap (12 .. 18)
                    if (m == 12) { ge.x[0] = (X3rGelu4::f2){o4[0], o4[1]}; ge.x[1] = (X3rGelu4::f2){o4[2], o4[3]}; ge.template stage<0>(); }
                    else if (m == 13) ge.template stage<1>();
                    else if (m == 14) ge.template stage<2>();
                    else if (m == 15) ge.template stage<3>();
                    else if (m == 16) ge.template stage<4>();
                    else if (m == 17) ge.template stage<5>();
                    else if (m == 18) { ge.template stage<6>(); o4 = (f32x4){ge.h[0], ge.h[1], ge.h[2], ge.h[3]}; }
                }
                if (s == 0 && pr == 0) X3R_MFMA0(acc[PAR][j], bf[pb][j], ap[s][pa]);
                else X3R_MFMA(acc[PAR][j], bf[pb][j], ap[s][pa]);
                // the next step's fragments into the registers whose last product of this step has been issued: weight plane 2 (last used by
                // product 1, m = 0 .. 3) in the slots of product 2 (m = 4 .. 7), plane 1 (product 4) in those of product 5 (16 .. 19), plane 0
                // right behind its own last MFMAs (20 .. 23); between them the copy of the step R_LEAD ahead
                const int rq = pr == 1 ? 2 : pr == 4 ? 1 : pr == 5 ? 0 : -1;
                if (rq >= 0) bf[rq][j] = *reinterpret_cast<const s16x8*>(nxt + rq * RB_PLANE + fb + j * 32 * 8);
                else if (m >= 8 && m < 11) { dma_piece((s + R_LEAD) & (R_RING - 1), m - 8); if (m == 10) dma_advance(); }
                else if (PAR == 1 && m == 11 && s < 4) { if (pf_rows > 0) prefetch(s); }
                __builtin_amdgcn_sched_barrier(0);
            }
            // the pending tile's quad of this step (one offset register for all sixteen quads: the rest is scalar)
            bstore_aux<X3R_STORE_AUX>(rsP, o4, crow0, (unsigned)(pend_n0 + ej * 32 + 8 * eg) * 4u);
            __builtin_amdgcn_sched_barrier(0);
            // step s + 2's weights have landed (the newer copies and the stores stay in flight; so do this wave's last fragment reads)
            X3R_T(0);
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(R_WAIT) : "memory");
            X3R_T(1);
#ifndef X3R_NOBARRIER
            asm volatile("s_barrier" ::: "memory");
#endif
            X3R_T(2);
        }
    };

    // Every unit has an EVEN number of tiles (launch_gemm_x3r), so the accumulator sets alternate statically: at a unit change set 1 is
    // the pending one and set 0 is dead - known to the register allocator, which would otherwise keep both alive across the fetch + split.
    for (;;) {
        const int m0 = panel * 128;
        for (int t = 0; t < nt; t += 2) {
            pf_rows = 0;
#ifndef X3R_NOPREFETCH
            if (t + 2 >= nt) {                                               // the unit's last pair: the next unit's panel, if there is one
                long long nx = pr_cur + nt / 2;
                if (nx >= pr_end) nx = (!in_tail && seg1_lo < seg1_hi) ? seg1_lo : -1;
                if (nx >= 0) {
                    const int mn = (int)(nx / un.ppp) * 128;
                    pf_base = p.A + (size_t)mn * p.lda;
                    pf_rows = p.M - mn < 128 ? p.M - mn : 128;
                }
            }
#endif
            tile(std::integral_constant<int, 0>{});
            pend_base = p.C + (size_t)m0 * p.ldc;
            pend_rows = p.M - m0 < 128 ? p.M - m0 : 128;
            pend_n0 = (t0 + t) * 128;
            tile(std::integral_constant<int, 1>{});
            pend_n0 = (t0 + t + 1) * 128;
        }
        pr_cur += nt / 2;
        if (pr_cur >= pr_end) {
            if (in_tail || seg1_lo >= seg1_hi) break;
            pr_cur = seg1_lo; pr_end = seg1_hi; in_tail = true;
        }
        front_unit(panel, t0, nt);
        X3R_T(5);
        load_a(panel);                  // (waits for its fetches, hence for every older copy and store)
        X3R_T(4);
    }
    // the last tile's stores
    {
        const __amdgpu_buffer_rsrc_t rsP = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(pend_base), 0, (int)((unsigned)pend_rows * (unsigned)p.ldc * 4u), 0x00020000);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 a4 = {acc[1][j][4 * g], acc[1][j][4 * g + 1], acc[1][j][4 * g + 2], acc[1][j][4 * g + 3]};
                f32x4 b4 = {0.f, 0.f, 0.f, 0.f};
                if (EPI) b4 = *reinterpret_cast<const f32x4*>(bias_sm + pend_n0 + j * 32 + 8 * g + 4 * hh);
                bstore_aux<X3R_STORE_AUX>(rsP, finish(a4, b4), crow0, (unsigned)(pend_n0 + j * 32 + 8 * g) * 4u);
            }
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
    if (p.K != RK || p.N % 256 != 0) return false;            // an even number of 128-column tiles (the accumulator sets alternate statically)
#ifndef X3R_STAMPS
    if (p.wsub) return false;
#endif
    if (p.gather || p.ksplit > 1 || p.residual || p.rowbias) return false;
    if (p.act > 1 || (p.act && !p.bias)) return false;      // epilogue instances: none, bias, bias + GELU
    if (p.bias && p.N > R_BIAS_MAX) return false;
    if ((p.ldc & 3) || (p.lda & 3)) return false;
    if (128ll * p.lda * 4 >= (1ll << 31) || 128ll * p.ldc * 4 >= (1ll << 31)) return false;
    return p.M >= 128 * 64;                       // a chip's worth of panels; smaller launches keep the tiled instances
}

hipError_t gemm_x3r_init() {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&mocha_gemm_x3r<-1>), hipFuncAttributeMaxDynamicSharedMemorySize, R_LDS_BYTES);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&mocha_gemm_x3r<0>), hipFuncAttributeMaxDynamicSharedMemorySize, R_LDS_BYTES);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&mocha_gemm_x3r<1>), hipFuncAttributeMaxDynamicSharedMemorySize, R_LDS_BYTES);
    return e;
}

hipError_t launch_gemm_x3r(const GemmParams& p, hipStream_t s, int grid) {
    if (!p.Wsplit || !gemm_x3r_supports(p)) return hipErrorInvalidValue;
    if (grid <= 0) grid = 256;
    const int panels = (p.M + 127) / 128;
    X3rUnits un;
    un.ppp = p.N / 256;
    un.pairs = (long long)panels * un.ppp;
    const int wgs = (int)std::min<long long>(grid, un.pairs);
#ifdef X3R_CONTIGUOUS
    un.whole = 0;                                   // one contiguous range per workgroup (measured 4 % slower on the 1 536-wide launch: every workgroup at another place of the weight image)
#else
    un.whole = panels / wgs;
#endif
    const dim3 g((unsigned)wgs);
    if (p.bias && p.act == 1) hipLaunchKernelGGL(mocha_gemm_x3r<1>, g, dim3(256), R_LDS_BYTES, s, p, un);
    else if (p.bias) hipLaunchKernelGGL(mocha_gemm_x3r<0>, g, dim3(256), R_LDS_BYTES, s, p, un);
    else hipLaunchKernelGGL(mocha_gemm_x3r<-1>, g, dim3(256), R_LDS_BYTES, s, p, un);
    return hipGetLastError();
}

}  // namespace mocha
