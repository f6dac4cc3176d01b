// Coarse pass of the many-query matcher against a bf16 bank, round 5:  S[z][q][n] = sum_{k in slice z} A[q][k] B[n][k]
// (A = centred bf16 query plane (Q, D), B = centred bf16 bank (N, D); the selection that follows is match_select2.hip / match_mfma.hip).
// The shape that matters is BASELINE configs[3]'s per-GPU share, 128 queries x 4 096 rows: 189 MB of bank stream through once and the
// arithmetic (24 GFLOP) is a third of the HBM time, so the kernel is built around BYTES IN FLIGHT, not around the matrix pipe.
//
// What round 4's mocha_match_gemm_bf16_dma did (match_mfma.hip): 128 x 128 tiles, both operands through one LDS-DMA ring of 4 x 32 KB -
// half of every stage is query bytes (L2 hits) and a CU has 48 KB of BANK bytes in flight: 3.6 TB/s on a cold bank.
// Here:
//   * 128 queries x 256 bank rows per 512-thread workgroup, K split 16 ways: query bytes per bank byte halve (ring traffic 1.1 MB per
//     workgroup instead of 1.47 MB).
//   * the BANK never touches LDS.  For v_mfma_f32_32x32x16_bf16 lane (r = lane & 31, h = lane >> 5) holds k = 8 h .. 8 h + 7 of row r:
//     16 contiguous bytes of the bank row in memory.  Every wave loads its own 32 rows' operands straight into registers
//     (global_load_dwordx4), PFS stages (64 k each) ahead: 8 waves x PFS x 4 KB = 160 KB of bank bytes in flight per CU at PFS = 5.
//   * only the queries go through the LDS-DMA ring (16 KB stages, PFS + 1 of them), swizzled as in match_mfma.hip: the 16-byte piece c of
//     query row r lives in slot c ^ ((r >> 1) & 7), conflict-free for the MFMA operand reads.
//   * one barrier per stage; every wave issues 2 DMA + 4 bank loads per stage in a fixed order, so ONE counted s_waitcnt per stage covers
//     both (vector-memory operations retire in order): at the top of stage s the PFS - 1 younger stages' 6 operations each may stay in flight.
//   * workgroups of one K slice share an XCD (blockIdx & 7), so the slice of the queries they all re-read lives in that XCD's L2.
// NT: the bank loads carry the non-temporal hint (read once; keeps the XCD's L2 for the query slices).
#include "kernels.h"
#include <mutex>
#include "device_utils.h"

namespace mocha {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

struct MatchPassParams {
    const unsigned short* A; const unsigned short* B; float* S;
    int Q; long long N; int D; int ksplit; long long slab_stride; int m_tiles, n_tiles;
    long long a_plane;            // elements between the two stacked query planes (NPL == 2)
    const unsigned short* Bt;     // TILED: the bank in operand order (mocha_tile32_bf16): [32-row tile][64-k stage][k16 step][lane] x 16 B
};

static constexpr int MP_Q = 128, MP_ROWS = 256, MP_BK = 64;

// K younger stages of OPS vector-memory operations each may stay in flight
template <int K, int OPS>
__device__ __forceinline__ void mp_wait_stages() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(K * OPS < 63 ? K * OPS : 63) : "memory"); }
// the same for a wave-uniform run-time k
template <int OPS>
__device__ __forceinline__ void mp_wait_dyn(int k) {
    switch (k) {
        case 0: mp_wait_stages<0, OPS>(); break;
        case 1: mp_wait_stages<1, OPS>(); break;
        case 2: mp_wait_stages<2, OPS>(); break;
        case 3: mp_wait_stages<3, OPS>(); break;
        case 4: mp_wait_stages<4, OPS>(); break;
        case 5: mp_wait_stages<5, OPS>(); break;
        case 6: mp_wait_stages<6, OPS>(); break;
        default: mp_wait_stages<7, OPS>(); break;
    }
}

// NPL = 2: the queries come as TWO stacked bf16 planes (a = a0 + a1 to 16 significant bits; the selection's error bound shrinks by 2^8 and
// with it the rows it must re-evaluate exactly: match_select2.hip).  The second plane doubles the ring's stages and the MFMAs, not the
// bank bytes - and here, unlike in round 4's kernel, the ring does not hold the bank.
// TILED: the bank operands come from an image in which every wave-level load is 1 KB of contiguous memory (row-major rows give 32 pieces
// of 32 B, 46 KB apart, per load)
template <int PFS, bool NT, bool FILL_ONLY, int NPL, bool TILED>
__global__ __launch_bounds__(512) void mocha_match_pass256(MatchPassParams p) {
    constexpr int R = PFS + 1;
    constexpr int OPS = 2 * NPL + 4;                                                  // vector-memory operations per wave and stage
    static_assert(PFS >= 2 && PFS <= 8 && (PFS - 1) * OPS <= 63, "the counted wait must fit vmcnt");
    extern __shared__ __attribute__((aligned(16))) unsigned short mp_sm[];          // [R][NPL][128 queries][64 k]
    constexpr int STAGE = NPL * MP_Q * MP_BK;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hh = lane >> 5;
    const int x = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int mt = j % p.m_tiles, pl = j / p.m_tiles;
    const int pp = pl * 8 + x;
    const int z = pp % p.ksplit, nt = pp / p.ksplit;
    if (nt >= p.n_tiles) return;
    const int m0 = mt * MP_Q;
    const long long n0 = (long long)nt * MP_ROWS;
    const int steps_total = p.D / MP_BK;
    const int per = (steps_total + p.ksplit - 1) / p.ksplit;
    const int s_begin = z * per;
    const int s_end = (s_begin + per) < steps_total ? (s_begin + per) : steps_total;
    const int nst = s_end > s_begin ? s_end - s_begin : 0;

    f32x16 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

    if (nst > 0) {
        const __amdgpu_buffer_rsrc_t rsA = make_rsrc(p.A + (size_t)m0 * p.D);
        const __amdgpu_buffer_rsrc_t rsA1 = make_rsrc(p.A + (size_t)(NPL == 2 ? p.a_plane : 0) + (size_t)m0 * p.D);
        unsigned a_off[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {                               // this wave's DMA pieces: 8-row pieces wave and wave + 8 of the 128 query rows
            const int r = 8 * (wave + 8 * i) + (lane >> 3);
            const int c = (lane & 7) ^ ((r >> 1) & 7);
            int ra = m0 + r; ra = ra < p.Q ? ra : p.Q - 1;
            a_off[i] = ((unsigned)(ra - m0) * (unsigned)p.D + c * 8u) * 2u;
        }
        long long rb = n0 + 32 * wave + l31; rb = rb < p.N ? rb : p.N - 1;
        const u32x4* bp = TILED ? reinterpret_cast<const u32x4*>(p.Bt) + ((size_t)(nt * 8 + wave) * steps_total + s_begin) * 256 + lane      // stage s, step ks: bp[(4 s + ks) 64]
                                : reinterpret_cast<const u32x4*>(p.B + (size_t)rb * p.D + (size_t)s_begin * MP_BK + 8 * hh);      // stage s, k16 step ks: bp[8 s + 2 ks]
        constexpr int BS = TILED ? 256 : 8, BK = TILED ? 64 : 2;
        unsigned ra_off[4], key_a[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) { const int r_a = i * 32 + l31; ra_off[i] = (unsigned)r_a * 64u; key_a[i] = (unsigned)((r_a >> 1) & 7); }

        auto issue = [&](int s, u32x4 (&breg)[4]) __attribute__((always_inline)) {     // 2 NPL DMA + 4 register loads, always in this order
            const unsigned so = (unsigned)((s_begin + s) * MP_BK) * 2u;
            unsigned short* st = mp_sm + (s % R) * STAGE;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (__attribute__((address_space(3))) void*)(st + (wave + 8 * i) * 512), 16, a_off[i], so, 0, 0);
                if (NPL == 2)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA1, (__attribute__((address_space(3))) void*)(st + MP_Q * MP_BK + (wave + 8 * i) * 512), 16, a_off[i], so, 0, 0);
            }
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) breg[ks] = NT ? __builtin_nontemporal_load(bp + BS * s + BK * ks) : bp[BS * s + BK * ks];
        };
        auto compute = [&](int s, const u32x4 (&breg)[4]) __attribute__((always_inline)) {
            const unsigned short* st = mp_sm + (s % R) * STAGE;
            if (FILL_ONLY) {                                        // floor measurement: the loads are kept alive, nothing is read from LDS or multiplied
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) acc[ks][0] += __uint_as_float((breg[ks][0] ^ breg[ks][1] ^ breg[ks][2] ^ breg[ks][3]) & 0x007fffffu);
                return;
            }
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                s16x8 a[4];
                const s16x8 b = __builtin_bit_cast(s16x8, breg[ks]);
                if (NPL == 2) {                                     // the low plane first: its products are 2^-8 of the high plane's
#pragma unroll
                    for (int i = 0; i < 4; ++i) a[i] = *reinterpret_cast<const s16x8*>(st + MP_Q * MP_BK + ra_off[i] + (((unsigned)(2 * ks + hh) ^ key_a[i]) << 3));
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, a[i], acc[i], 0, 0, 0);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) a[i] = *reinterpret_cast<const s16x8*>(st + ra_off[i] + (((unsigned)(2 * ks + hh) ^ key_a[i]) << 3));
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, a[i], acc[i], 0, 0, 0);      // C^T tile: lane = query
            }
        };

        u32x4 breg[PFS][4];
        int s = 0;
        if (nst >= 2 * PFS) {
            // steady state: an UNCONDITIONAL prologue (the compiler's own wait-count analysis must see PFS - 1 younger stages behind every
            // register set on every path into the loop, or it drains the queue at the first use), then whole groups of PFS stages that all issue
            // another one (static register indices, fixed operation counts)
#pragma unroll
            for (int u = 0; u < PFS; ++u) issue(u, breg[u]);
            for (; s + 2 * PFS <= nst; s += PFS) {
#pragma unroll
                for (int u = 0; u < PFS; ++u) {
                    mp_wait_stages<PFS - 1, OPS>();
                    __builtin_amdgcn_s_barrier();
                    compute(s + u, breg[u]);
                    // the slot stage s + u + PFS goes into is the one stage s + u - 1 was read from: every wave is past this stage's barrier
                    issue(s + u + PFS, breg[u]);
                }
            }
        } else {
#pragma unroll
            for (int u = 0; u < PFS; ++u)
                if (u < nst) issue(u, breg[u]);
        }
        // tail: between PFS and 2 PFS - 1 stages left (or fewer than PFS in all); the first of them still issue
        const int rem = nst - s;
#pragma unroll
        for (int u = 0; u < 2 * PFS - 1; ++u) {
            if (u < rem) {
                const int younger = (rem - 1 - u) < (PFS - 1) ? (rem - 1 - u) : (PFS - 1);
                mp_wait_dyn<OPS>(younger);
                __builtin_amdgcn_s_barrier();
                compute(s + u, breg[u % PFS]);
                if (u + PFS < rem) issue(s + u + PFS, breg[u % PFS]);
            }
        }
    }

    float* Sz = p.S + (size_t)z * p.slab_stride;
    const bool vec = (p.N & 3) == 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = m0 + i * 32 + l31;
        if (row >= p.Q) continue;
        float* srow = Sz + (size_t)row * p.N;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const long long col = n0 + 32 * wave + 8 * g + 4 * hh;
            if (col >= p.N) continue;
            if (vec) {
                const f32x4 v = {acc[i][4 * g], acc[i][4 * g + 1], acc[i][4 * g + 2], acc[i][4 * g + 3]};
                *reinterpret_cast<f32x4*>(srow + col) = v;
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (col + e < p.N) srow[col + e] = acc[i][4 * g + e];
            }
        }
    }
}

template <int PFS, int NPL>
static constexpr size_t mp_lds_bytes() { return (size_t)(PFS + 1) * NPL * MP_Q * MP_BK * sizeof(unsigned short); }

// K split of the 256-row pass: a power of two <= 16 with about one workgroup per CU
int match_pass256_ksplit(int Q, int64_t N) {
    const long long tiles = (long long)((Q + MP_Q - 1) / MP_Q) * ((N + MP_ROWS - 1) / MP_ROWS);
    int k = 16;
    while (k > 1 && tiles * k > 320) k >>= 1;
    return k;
}

template <int PFS, bool NT, bool FILL, int NPL, bool TILED>
static hipError_t mp_launch2(const MatchPassParams& p, unsigned grid, hipStream_t s) {
    // once per instantiation, whichever host thread comes first (contexts may be driven from several threads)
    static std::once_flag once;
    static hipError_t attr_err = hipSuccess;
    std::call_once(once, [] {
        attr_err = hipFuncSetAttribute(reinterpret_cast<const void*>(&mocha_match_pass256<PFS, NT, FILL, NPL, TILED>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)mp_lds_bytes<PFS, NPL>());
    });
    if (attr_err != hipSuccess) return attr_err;
    hipLaunchKernelGGL((mocha_match_pass256<PFS, NT, FILL, NPL, TILED>), dim3(grid), dim3(512), (mp_lds_bytes<PFS, NPL>()), s, p);
    return hipGetLastError();
}
template <int PFS, bool NT, bool FILL, int NPL>
static hipError_t mp_launch(const MatchPassParams& p, unsigned grid, hipStream_t s) {
    return p.Bt ? mp_launch2<PFS, NT, FILL, NPL, true>(p, grid, s) : mp_launch2<PFS, NT, FILL, NPL, false>(p, grid, s);
}

// bank16 (N, D) bf16 row-major -> the pass's operand-order image: block (32-row tile, 64-k stage) = 4 k16 steps x 64 lanes x 16 B; rows past N
// repeat the last row.  One 256-thread workgroup per block.
__global__ __launch_bounds__(256) void mocha_tile32_bf16(const unsigned short* __restrict__ bank16, unsigned short* __restrict__ out, long long N, int D) {
    const int stages = D / MP_BK;
    const long long rt = blockIdx.x / stages; const int st = (int)(blockIdx.x - rt * stages);
    const int ks = threadIdx.x >> 6, lane = threadIdx.x & 63;
    long long row = rt * 32 + (lane & 31); row = row < N ? row : N - 1;
    const u32x4 v = *reinterpret_cast<const u32x4*>(bank16 + (size_t)row * D + st * MP_BK + ks * 16 + 8 * (lane >> 5));
    reinterpret_cast<u32x4*>(out)[(size_t)blockIdx.x * 256 + threadIdx.x] = v;
}
size_t match_tile32_elems(int64_t N, int D) { return (size_t)((N + 255) / 256) * 256 * (size_t)D; }       // whole 256-row workgroup tiles
hipError_t launch_tile32_bf16(const void* bank16, void* out, int64_t N, int D, hipStream_t s) {
    if (N <= 0) return hipSuccess;
    if (D % MP_BK) return hipErrorInvalidValue;
    const long long blocks = (long long)((N + 255) / 256) * 8 * (D / MP_BK);
    if (blocks > 0x7fffffffll) return hipErrorInvalidValue;
    hipLaunchKernelGGL(mocha_tile32_bf16, dim3((unsigned)blocks), dim3(256), 0, s, (const unsigned short*)bank16, (unsigned short*)out, (long long)N, D);
    return hipGetLastError();
}

// variant: bits 0-3 = PFS (one plane 3 .. 6, default 5; two planes 2 .. 4, default 3), bit 4 = non-temporal bank loads, bit 8 = fill only
// (measurement: S is garbage).  planes = 2: qc16 holds two stacked planes, plane 1 starts Q * D elements after plane 0.
hipError_t launch_match_pass256(const void* qc16, const void* bank16, float* S, int Q, int64_t N, int D, int ksplit, hipStream_t s, int variant, int planes,
                                const void* tiled32) {
    if (Q <= 0 || N <= 0) return hipSuccess;
    if (D % MP_BK || ksplit < 1 || ksplit > 16 || (ksplit & (ksplit - 1)) || (planes != 1 && planes != 2)) return hipErrorInvalidValue;
    if ((long long)MP_Q * D * 2 >= (1ll << 31)) return hipErrorInvalidValue;              // 32-bit buffer offsets inside a query tile
    MatchPassParams p;
    p.A = (const unsigned short*)qc16; p.B = (const unsigned short*)bank16; p.S = S;
    p.Q = Q; p.N = N; p.D = D; p.ksplit = ksplit; p.slab_stride = (long long)Q * N;
    p.m_tiles = (Q + MP_Q - 1) / MP_Q; p.n_tiles = (int)((N + MP_ROWS - 1) / MP_ROWS);
    p.a_plane = (long long)Q * D;
    p.Bt = (const unsigned short*)tiled32;
    const long long pairs = (long long)p.n_tiles * ksplit;
    const unsigned grid = (unsigned)(((pairs + 7) / 8) * p.m_tiles * 8);
    const int pfs = (variant & 15) ? (variant & 15) : (planes == 2 ? 3 : 5);
    const bool nt = (variant & 16) != 0, fill = (variant & 256) != 0;
#define MP_CASE(P, NPL)                                                                                          \
    case P:                                                                                                      \
        if (fill) return nt ? mp_launch<P, true, true, NPL>(p, grid, s) : mp_launch<P, false, true, NPL>(p, grid, s);   \
        return nt ? mp_launch<P, true, false, NPL>(p, grid, s) : mp_launch<P, false, false, NPL>(p, grid, s);
    if (planes == 2) {
        switch (pfs) {
            MP_CASE(2, 2) MP_CASE(3, 2) MP_CASE(4, 2)
            default: return hipErrorInvalidValue;
        }
    }
    switch (pfs) {
        MP_CASE(3, 1) MP_CASE(4, 1) MP_CASE(5, 1) MP_CASE(6, 1)
        default: return hipErrorInvalidValue;
    }
#undef MP_CASE
}

}  // namespace mocha
