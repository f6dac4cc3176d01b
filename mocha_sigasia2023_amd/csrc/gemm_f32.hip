// fp32 MFMA GEMM for gfx950 (CDNA4): C = epilogue(Aop · W^T), exact-f32 v_mfma_f32_32x32x2_f32.
//
// Why f32 MFMA: the path must match the reference's fp32 CPU result to 1e-4 through ~25 chained
// contractions with no normalisation layers in between; v_mfma_f32_32x32x2_f32 is a k-ordered
// fmaf chain (bit-for-bit f32) at the full f32 rate (157 TFLOP/s dense peak on MI355X).
//
// Structure (measured reasons in DESIGN_HISTORY.md §5):
//   * 128 x BN x 32 tile per 256-thread workgroup (4 waves, each TM x TN MFMA tiles of 32x32).  One
//     LDS stage (36.9 KB for BN = 128) so that three workgroups share a CU: the K loops of this path
//     are short (6-40 slabs), and a third co-resident workgroup hides more of the prologue /
//     epilogue of its neighbours than a second LDS stage hides inside one workgroup (measured).
//     The next slab is prefetched global->VGPR while the current one is multiplied.
//   * LDS rows are 36 floats (32 + 4 pad): the per-lane ds_read_b128 of four consecutive k is
//     bank-conflict free for every 16-lane service group (36*i mod 64 = 16 distinct multiples of 4).
//     Each lane's four k values feed four consecutive MFMAs; A and B use the same k permutation.
//   * The MFMA is issued with the operands swapped (W tile as "A", activation tile as "B"), so an
//     accumulator holds C^T: lane&31 = output row, registers 4g..4g+3 = four consecutive output
//     columns -> 16-byte stores and 16-byte bias / residual fetches (4x fewer store instructions).
//   * XCD-aware tile order: the n-tiles of one m-tile get block ids that are equal mod 8, so they
//     run on one XCD and share its L2 copy of the A tile (placement is a speed matter only).
//
// The A operand can be gathered on the fly (temporal conv as GEMM, reflect padding, folded
// average pooling / nearest upsampling), see kernels.h.
#include "kernels.h"
#include "device_utils.h"

namespace mocha {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

static constexpr int BM = 128;
static constexpr int BK = 32;
static constexpr int LDSK = BK + 4;

__device__ __forceinline__ float lrelu02(float x) { return x > 0.f ? x : 0.2f * x; }
__device__ __forceinline__ float gelu_erf(float x) { return mocha_gelu(x); }

template <int BN, int WM, int WN, int TM, int TN>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(BN == 64 ? 5 : 3))) void mocha_gemm_f32(GemmParams p) {
    constexpr int BM = WM * TM * 32;                // rows of the tile: 128, or 64 for the mid-size instance <64,2,2,1,1>
    static_assert(WM * WN == 4 && WN * TN * 32 == BN, "tile shape");
    constexpr int NA = BM / 32;                     // float4 loads of A per thread per slab
    constexpr int NB = BN / 32;                     // float4 loads of W per thread per slab
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                               // [BM][LDSK]
    float* Bs = smem + BM * LDSK;                   // [BN][LDSK]

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hh = lane >> 5;
    const int wm = wave / WN, wn = wave % WN;

    const int n_tiles = (p.N + BN - 1) / BN;
    const int m_tiles = (p.M + BM - 1) / BM;
    const int bid = blockIdx.x;
    int mt, nt;
    if (m_tiles >= 8) {                 // XCD-aware order: the n-tiles of one m-tile share an XCD (bid % 8)
        const int grp = bid / (8 * n_tiles);
        const int rem = bid - grp * 8 * n_tiles;
        mt = grp * 8 + (rem & 7);
        nt = rem >> 3;
    } else {                            // few row tiles (single windows, small query sets): spread over all XCDs
        mt = bid / n_tiles;
        nt = bid - mt * n_tiles;
    }
    if (mt >= m_tiles) return;
    const int m0 = mt * BM, n0 = nt * BN;

    const int slabs_total = p.K / BK;               // K % 32 == 0 is checked on the host
    const int per = (slabs_total + p.ksplit - 1) / p.ksplit;
    const int s_begin = blockIdx.z * per;
    const int s_end = (s_begin + per) < slabs_total ? (s_begin + per) : slabs_total;

    // ---- loader: 8 threads cover one 128-byte row segment of a slab
    const int lrow = tid >> 3;
    const int lcol = (tid & 7) * 4;
    // plain A: the resource starts at the tile's first row, lane offsets are (row - m0) * lda + lcol (always < 2 GiB);
    // gathered A: the resource is the whole source tensor (checked < 2 GiB on the host), lane offsets are recomputed per tap;
    // W: the resource starts at the tile's first weight row.
    int a_rb[NA], a_t[NA];
    unsigned a_off[NA], w_off[NB];
    const __amdgpu_buffer_rsrc_t rsA = make_rsrc(p.gather ? p.A : p.A + (size_t)(m0 < p.M ? m0 : 0) * p.lda);
    const __amdgpu_buffer_rsrc_t rsW = make_rsrc(p.W + (size_t)n0 * p.K);
    {
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            int m = m0 + lrow + 32 * i;
            m = m < p.M ? m : p.M - 1;
            if (p.gather) {
                const int v = m % p.V;
                const int bt = m / p.V;
                const int t = bt % p.T_out;
                const int b = bt / p.T_out;
                a_rb[i] = b * p.T_src * p.V + v;
                a_t[i] = t;
                a_off[i] = 0;
            } else {
                a_rb[i] = m;
                a_t[i] = 0;
                a_off[i] = ((unsigned)(m - m0) * (unsigned)p.lda + lcol) * 4u;
            }
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            int n = n0 + lrow + 32 * i;
            n = n < p.N ? n : p.N - 1;
            w_off[i] = ((unsigned)(n - n0) * (unsigned)p.K + lcol) * 4u;
        }
    }

    f32x4 ra[NA], rb[NB];
    auto load_slab = [&](int s) __attribute__((always_inline)) {
        const int k0 = s * BK;
        if (!p.gather) {
#pragma unroll
            for (int i = 0; i < NA; ++i) ra[i] = bload(rsA, a_off[i], (unsigned)k0 * 4u);
        } else if (p.R == 1) {
            // a slab stays inside one tap for Cc / 32 slabs: the row offsets of the tap are computed when it starts
            const int tap = k0 / p.Cc;
            const int cin = k0 - tap * p.Cc;
            if (cin == 0 || s == s_begin) {
#pragma unroll
                for (int i = 0; i < NA; ++i) {
                    int tf = a_t[i] * p.stride + tap * p.tstep - p.pad;
                    tf = tf < 0 ? -tf : tf;
                    tf = tf >= p.T_full ? 2 * (p.T_full - 1) - tf : tf;
                    a_off[i] = ((unsigned)(a_rb[i] + (tf >> p.tshift) * p.V) * (unsigned)p.lda + lcol) * 4u;
                }
            }
#pragma unroll
            for (int i = 0; i < NA; ++i) ra[i] = bload(rsA, a_off[i], (unsigned)cin * 4u);
        } else {
            const int tap = k0 / p.Cc;
            const unsigned c0 = (unsigned)(k0 - tap * p.Cc + lcol) * 4u;
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                f32x4 acc4 = {0.f, 0.f, 0.f, 0.f};
                for (int j = 0; j < p.R; ++j) {
                    int tf = a_t[i] * p.stride + j + tap * p.tstep - p.pad;
                    tf = tf < 0 ? -tf : tf;
                    tf = tf >= p.T_full ? 2 * (p.T_full - 1) - tf : tf;
                    const int row = a_rb[i] + (tf >> p.tshift) * p.V;
                    acc4 += bload(rsA, (unsigned)row * (unsigned)p.lda * 4u + c0, 0u);
                }
                ra[i] = acc4 * p.ascale;
            }
        }
        if (p.a_lrelu) {
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                ra[i][0] = lrelu02(ra[i][0]); ra[i][1] = lrelu02(ra[i][1]);
                ra[i][2] = lrelu02(ra[i][2]); ra[i][3] = lrelu02(ra[i][3]);
            }
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) rb[i] = bload(rsW, w_off[i], (unsigned)k0 * 4u);
        if (p.wsub) {                               // matcher: bank rows are centred on the fly (the bank itself may be borrowed)
            const f32x4 cv = *reinterpret_cast<const f32x4*>(p.wsub + k0 + lcol);
#pragma unroll
            for (int i = 0; i < NB; ++i) rb[i] -= cv;
        }
    };
    auto store_slab = [&]() __attribute__((always_inline)) {
        float* Ab = As;
        float* Bb = Bs;
#pragma unroll
        for (int i = 0; i < NA; ++i) *reinterpret_cast<f32x4*>(Ab + (lrow + 32 * i) * LDSK + lcol) = ra[i];
#pragma unroll
        for (int i = 0; i < NB; ++i) *reinterpret_cast<f32x4*>(Bb + (lrow + 32 * i) * LDSK + lcol) = rb[i];
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    if (s_begin < s_end) {
        load_slab(s_begin);
        store_slab();
    }
    __syncthreads();

    const float* Ab = As + (wm * TM * 32 + l31) * LDSK + 4 * hh;
    const float* Bb = Bs + (wn * TN * 32 + l31) * LDSK + 4 * hh;
    for (int s = s_begin; s < s_end; ++s) {
        const bool more = (s + 1) < s_end;
        if (more) load_slab(s + 1);                 // global -> VGPR, in flight under the MFMAs
#pragma unroll
        for (int kg = 0; kg < BK / 8; ++kg) {
            f32x4 a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = *reinterpret_cast<const f32x4*>(Ab + i * 32 * LDSK + kg * 8);
#pragma unroll
            for (int j = 0; j < TN; ++j) b[j] = *reinterpret_cast<const f32x4*>(Bb + j * 32 * LDSK + kg * 8);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(b[j][ks], a[i][ks], acc[i][j], 0, 0, 0);   // C^T tile
        }
        __syncthreads();                            // every wave is done reading the stage
        if (more) store_slab();
        __syncthreads();
    }

    // ---- epilogue: acc[i][j] holds C^T of MFMA tile (i, j): lane&31 = row, regs 4g..4g+3 = 4 columns.
    // Every access goes through a buffer resource anchored at the tile (SGPR base), so a lane needs one 32-bit row offset per
    // row block and the (j, g) column steps are instruction immediates: no 64-bit address arithmetic next to the
    // co-resident workgroups' MFMAs.
    float* Cz = p.C + (size_t)blockIdx.z * p.slab_stride;
    const bool raw = p.ksplit > 1;
    const bool vec_ok = ((p.ldc & 3) == 0) && (!p.residual || (p.ldr & 3) == 0) && ((p.N & 3) == 0 || raw);
    const __amdgpu_buffer_rsrc_t rsC = make_rsrc(Cz + (size_t)m0 * p.ldc + n0);
    const __amdgpu_buffer_rsrc_t rsBias = make_rsrc((!raw && p.bias) ? p.bias + n0 : p.W);
    const __amdgpu_buffer_rsrc_t rsRb = make_rsrc((!raw && p.rowbias) ? p.rowbias + n0 : p.W);
    const __amdgpu_buffer_rsrc_t rsRes = make_rsrc((!raw && p.residual) ? p.residual + (size_t)m0 * p.ldr + n0 : p.W);
    const unsigned colb = (unsigned)(wn * TN * 32 + 4 * hh) * 4u;            // this lane's first column in the tile, bytes

    if (vec_ok && n0 + BN <= p.N) {
        // Full-width tile: the accumulators (lane = row, 4 columns per register group) are transposed through LDS, 64 rows
        // at a time, so that every store instruction writes whole 128-byte lines and bias / residual are read the same way.
        // Measured with PMC on M = 105 300, N = 256: storing straight from the C^T layout (32 contiguous bytes per row and
        // instruction) wrote 150-170 MB to HBM for a 108 MB result.
        constexpr int LDP = BN + 4;                  // row pitch in floats: 8 consecutive rows cover all banks for b128 accesses
        constexpr int C4 = BN / 4;                   // float4 per row
        static_assert(64 * LDP <= (BM + BN) * LDSK, "epilogue staging fits the operand stages");
        static_assert(BM % 64 == 0, "staged 64 rows at a time");
        float* stage = smem;
#pragma unroll
        for (int h = 0; h < BM / 64; ++h) {
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int rblk = wm * TM + i;        // 32-row block of the tile
                if ((rblk >> 1) != h) continue;      // compile-time per wave position: uniform
                float* srow = stage + ((rblk & 1) * 32 + l31) * LDP + wn * TN * 32 + 4 * hh;
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        f32x4 v = {acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
                        *reinterpret_cast<f32x4*>(srow + j * 32 + 8 * g) = v;
                    }
            }
            __syncthreads();
#pragma unroll
            for (int it = 0; it < 64 * C4 / 256; ++it) {
                const int e = tid + 256 * it;
                const int r = e / C4, c4 = e - r * C4;
                const int rloc = 64 * h + r;
                const int row = m0 + rloc;
                if (row < p.M) {
                    f32x4 v = *reinterpret_cast<const f32x4*>(stage + r * LDP + c4 * 4);
                    const unsigned cb = (unsigned)c4 * 16u;
                    if (!raw) {
                        if (p.bias) v += bload(rsBias, cb, 0u);
                        if (p.rowbias) v += bload(rsRb, (unsigned)(row % p.rb_mod) * (unsigned)p.N * 4u + cb, 0u);
                        if (p.act == 1) { v[0] = gelu_erf(v[0]); v[1] = gelu_erf(v[1]); v[2] = gelu_erf(v[2]); v[3] = gelu_erf(v[3]); }
                        else if (p.act == 2) { v[0] = lrelu02(v[0]); v[1] = lrelu02(v[1]); v[2] = lrelu02(v[2]); v[3] = lrelu02(v[3]); }
                        else if (p.act == 3) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f); }
                        if (p.residual) v += bload(rsRes, (unsigned)rloc * (unsigned)p.ldr * 4u + cb, 0u);
                    }
                    bstore(rsC, v, (unsigned)rloc * (unsigned)p.ldc * 4u + cb, 0u);
                }
            }
            if (h + 1 < BM / 64) __syncthreads();
        }
        return;
    }

    // ragged tiles (the matcher's split-K slabs, N not a multiple of the tile): straight from the accumulators
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int rloc = (wm * TM + i) * 32 + l31;
        const int row = m0 + rloc;
        if (row >= p.M) continue;
        const float* rbrow = (!raw && p.rowbias) ? p.rowbias + (size_t)(row % p.rb_mod) * p.N : nullptr;
        const float* rsrow = (!raw && p.residual) ? p.residual + (size_t)row * p.ldr : nullptr;
        float* crow = Cz + (size_t)row * p.ldc;
        const unsigned c_off = (unsigned)rloc * (unsigned)p.ldc * 4u + colb;
        const unsigned rb_off = rbrow ? (unsigned)(row % p.rb_mod) * (unsigned)p.N * 4u + colb : 0u;
        const unsigned rs_off = rsrow ? (unsigned)rloc * (unsigned)p.ldr * 4u + colb : 0u;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int col = n0 + (wn * TN + j) * 32 + 8 * g + 4 * hh;
                if (col >= p.N) continue;
                const unsigned cstep = (unsigned)(j * 32 + 8 * g) * 4u;       // compile-time: folds into the instruction offset
                f32x4 v = {acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
                if (vec_ok && col + 3 < p.N) {
                    if (!raw) {
                        if (p.bias) v += bload(rsBias, colb + cstep, 0u);
                        if (rbrow) v += bload(rsRb, rb_off + cstep, 0u);
                        if (p.act == 1) { v[0] = gelu_erf(v[0]); v[1] = gelu_erf(v[1]); v[2] = gelu_erf(v[2]); v[3] = gelu_erf(v[3]); }
                        else if (p.act == 2) { v[0] = lrelu02(v[0]); v[1] = lrelu02(v[1]); v[2] = lrelu02(v[2]); v[3] = lrelu02(v[3]); }
                        else if (p.act == 3) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f); }
                        if (rsrow) v += bload(rsRes, rs_off + cstep, 0u);
                    }
                    bstore(rsC, v, c_off + cstep, 0u);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int c1 = col + e;
                        if (c1 >= p.N) continue;
                        float x = v[e];
                        if (!raw) {
                            if (p.bias) x += p.bias[c1];
                            if (rbrow) x += rbrow[c1];
                            if (p.act == 1) x = gelu_erf(x);
                            else if (p.act == 2) x = lrelu02(x);
                            else if (p.act == 3) x = fmaxf(x, 0.f);
                            if (rsrow) x += rsrow[c1];
                        }
                        crow[c1] = x;
                    }
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------
// Skinny variant for a handful of windows (streaming one window per step: M = 90 ... 1440).
// The tiled kernel above would launch 2-24 workgroups with a serial K loop of up to 40 slabs; here
// a workgroup owns one 32x32 output tile, its 4 waves split K four ways, every operand goes
// global -> register in MFMA layout (no LDS staging, all loads of a wave in flight at once), and the
// four partial tiles are summed in a fixed order through LDS (deterministic) before the same
// fused epilogue.  Latency of a GEMM drops from ~K/32 slab times to ~K/128 MFMA groups + one load.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void mocha_gemm_skinny(GemmParams p) {
    __shared__ float red[3][16][64];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hh = lane >> 5;
    const int n_tiles = (p.N + 31) / 32;
    const int mt = blockIdx.x / n_tiles, nt = blockIdx.x - mt * n_tiles;
    const int m0 = mt * 32, n0 = nt * 32;

    int m = m0 + l31;
    m = m < p.M ? m : p.M - 1;
    int a_rb = m, a_t = 0;
    if (p.gather) {
        const int v = m % p.V;
        const int bt = m / p.V;
        a_t = bt % p.T_out;
        a_rb = (bt / p.T_out) * p.T_src * p.V + v;
    }
    int n = n0 + l31;
    n = n < p.N ? n : p.N - 1;
    // buffer loads (device_utils.h): tile-anchored resources, 32-bit lane offsets, the k position as the scalar offset
    const __amdgpu_buffer_rsrc_t rsA = make_rsrc(p.gather ? p.A : p.A + (size_t)(m0 < p.M ? m0 : 0) * p.lda);
    const __amdgpu_buffer_rsrc_t rsW = make_rsrc(p.W + (size_t)n0 * p.K);
    const unsigned a_off = p.gather ? 0u : ((unsigned)(m - m0) * (unsigned)p.lda + 4u * hh) * 4u;
    const unsigned w_off = ((unsigned)(n - n0) * (unsigned)p.K + 4u * hh) * 4u;

    const int groups = p.K / 8;                     // k groups of 8 (4 per lane half)
    const int gper = (groups + 3) / 4;
    const int g_begin = wave * gper;
    const int g_end = (g_begin + gper) < groups ? (g_begin + gper) : groups;

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;

    // The kernel is a latency chain (load round, 64 dependent MFMAs, reduction, epilogue): the epilogue's operands are
    // fetched by the wave that will use them before the K loop, so their round trip is hidden under it.
    const int row = m0 + l31;
    f32x4 ebias[4], eres[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        f32x4 z = {0.f, 0.f, 0.f, 0.f};
        ebias[g] = z; eres[g] = z;
        const int col = n0 + 8 * g + 4 * hh;
        if (wave == 0 && row < p.M && col < p.N) {
            if (p.bias) ebias[g] = *reinterpret_cast<const f32x4*>(p.bias + col);
            if (p.rowbias) ebias[g] += *reinterpret_cast<const f32x4*>(p.rowbias + (size_t)(row % p.rb_mod) * p.N + col);
            if (p.residual) eres[g] = *reinterpret_cast<const f32x4*>(p.residual + (size_t)row * p.ldr + col);
        }
    }

    for (int g0 = g_begin; g0 < g_end; g0 += 8) {
        f32x4 a4[8], b4[8];
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            const int kb = (g0 + g) * 8;             // wave-uniform
            f32x4 z = {0.f, 0.f, 0.f, 0.f};
            a4[g] = z; b4[g] = z;
            if (g0 + g < g_end) {
                if (!p.gather) {
                    a4[g] = bload(rsA, a_off, (unsigned)kb * 4u);
                } else {
                    const int k = kb + 4 * hh;
                    const int tap = k / p.Cc;
                    const int cc = k - tap * p.Cc;
                    int tf = a_t * p.stride + tap * p.tstep - p.pad;
                    tf = tf < 0 ? -tf : tf;
                    tf = tf >= p.T_full ? 2 * (p.T_full - 1) - tf : tf;
                    a4[g] = bload(rsA, ((unsigned)(a_rb + (tf >> p.tshift) * p.V) * (unsigned)p.lda + (unsigned)cc) * 4u, 0u);
                }
                b4[g] = bload(rsW, w_off, (unsigned)kb * 4u);
            }
        }
        if (p.a_lrelu) {
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                a4[g][0] = lrelu02(a4[g][0]); a4[g][1] = lrelu02(a4[g][1]);
                a4[g][2] = lrelu02(a4[g][2]); a4[g][3] = lrelu02(a4[g][3]);
            }
        }
#pragma unroll
        for (int g = 0; g < 8; ++g)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(b4[g][ks], a4[g][ks], acc, 0, 0, 0);      // C^T tile; zero groups add 0
    }
    if (wave > 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) red[wave - 1][r][lane] = acc[r];
    }
    __syncthreads();
    if (wave > 0) return;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = ((acc[r] + red[0][r][lane]) + red[1][r][lane]) + red[2][r][lane];

    if (row >= p.M) return;
    float* crow = p.C + (size_t)row * p.ldc;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int col = n0 + 8 * g + 4 * hh;
        if (col >= p.N) continue;                   // N % 4 == 0 is checked on the host for this kernel
        f32x4 v = {acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]};
        v += ebias[g];
        if (p.act == 1) { v[0] = gelu_erf(v[0]); v[1] = gelu_erf(v[1]); v[2] = gelu_erf(v[2]); v[3] = gelu_erf(v[3]); }
        else if (p.act == 2) { v[0] = lrelu02(v[0]); v[1] = lrelu02(v[1]); v[2] = lrelu02(v[2]); v[3] = lrelu02(v[3]); }
        else if (p.act == 3) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f); }
        v += eres[g];
        *reinterpret_cast<f32x4*>(crow + col) = v;
    }
}

// true when the tiled kernel would leave most of the chip idle and the shape fits the skinny kernel
bool gemm_is_skinny(const GemmParams& p) {
    if (p.wsub || p.ksplit > 1 || (p.N & 3) || (p.ldc & 3) || (p.residual && (p.ldr & 3)) || (p.gather && p.R != 1)) return false;
    const long long wide_tiles = (long long)((p.M + BM - 1) / BM) * ((p.N + 63) / 64);
    return wide_tiles < 96;
}

// ---------------------------------------------------------------------------------------
// A handful of windows (M <= 768: the streamed per-window step, the CVAE branch for up to 4-8 clips).  The 32 x 32 variant above puts such a
// GEMM on 24 ... 144 workgroups - a tenth of the chip's SIMDs - and a wave walks its K quarter in rounds of 64 (load round trip,
// 32 dependent 64-cycle MFMAs, repeat): 8-16 us per launch, 30 launches per window.  Here a workgroup owns a 16 x 16 tile
// (v_mfma_f32_16x16x4_f32, 32 cycles), so the same GEMM spreads over 4x the workgroups, and a wave issues EVERY load of its K
// quarter (up to 16 + 16 sixteen-byte loads per lane) before the first MFMA: one memory round trip, then at most 64 MFMAs on two
// alternating accumulators.  Same operands-in-MFMA-layout loads, fixed-order reduction of the four K quarters through LDS
// (deterministic) and fused epilogue as above.
//   lane l: A row m0 + (l & 15), W row n0 + (l & 15), k sub-block l >> 4 (4 consecutive k per 16-byte load: element i feeds MFMA i of
//   the group, the same k permutation on both operands);  acc[r] = C[m0 + (l & 15)][n0 + 4 (l >> 4) + r]  (W as the "A" operand).
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void mocha_gemm_skinny16(GemmParams p) {
    __shared__ f32x4 red[3][64];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, kq = lane >> 4;
    const int n_tiles = (p.N + 15) / 16;
    const int mt = blockIdx.x / n_tiles, nt = blockIdx.x - mt * n_tiles;
    const int m0 = mt * 16, n0 = nt * 16;

    int m = m0 + l15;
    m = m < p.M ? m : p.M - 1;
    int a_rb = m, a_t = 0;
    if (p.gather) {
        const int v = m % p.V;
        const int bt = m / p.V;
        a_t = bt % p.T_out;
        a_rb = (bt / p.T_out) * p.T_src * p.V + v;
    }
    int n = n0 + l15;
    n = n < p.N ? n : p.N - 1;
    const __amdgpu_buffer_rsrc_t rsA = make_rsrc(p.gather ? p.A : p.A + (size_t)(m0 < p.M ? m0 : 0) * p.lda);
    const __amdgpu_buffer_rsrc_t rsW = make_rsrc(p.W + (size_t)n0 * p.K);
    const unsigned a_off = p.gather ? 0u : ((unsigned)(m - m0) * (unsigned)p.lda + 4u * kq) * 4u;
    const unsigned w_off = ((unsigned)(n - n0) * (unsigned)p.K + 4u * kq) * 4u;

    const int groups = p.K / 16;                    // k groups of 16 (4 per lane quarter)
    const int gper = (groups + 3) / 4;
    const int g_begin = wave * gper;
    const int g_end = (g_begin + gper) < groups ? (g_begin + gper) : groups;

    // epilogue operands of the wave that will use them, fetched before the K loop (their round trip hides under it)
    const int row = m0 + l15, col = n0 + 4 * kq;
    f32x4 ebias = {0.f, 0.f, 0.f, 0.f}, eres = {0.f, 0.f, 0.f, 0.f};
    if (wave == 0 && row < p.M && col < p.N) {
        if (p.bias) ebias = *reinterpret_cast<const f32x4*>(p.bias + col);
        if (p.rowbias) ebias += *reinterpret_cast<const f32x4*>(p.rowbias + (size_t)(row % p.rb_mod) * p.N + col);
        if (p.residual) eres = *reinterpret_cast<const f32x4*>(p.residual + (size_t)row * p.ldr + col);
    }

    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    constexpr int GMAX = 16;                        // groups in flight per round: 2 x 16 x 16 B per lane (128 VGPRs)
    for (int g0 = g_begin; g0 < g_end; g0 += GMAX) {
        f32x4 a4[GMAX], b4[GMAX];
#pragma unroll
        for (int g = 0; g < GMAX; ++g) {
            const int kb = (g0 + g) * 16;            // wave-uniform
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
            a4[g] = z; b4[g] = z;
            if (g0 + g < g_end) {
                if (!p.gather) {
                    a4[g] = bload(rsA, a_off, (unsigned)kb * 4u);
                } else {
                    const int k = kb + 4 * kq;
                    const int tap = k / p.Cc;
                    const int cc = k - tap * p.Cc;
                    int tf = a_t * p.stride + tap * p.tstep - p.pad;
                    tf = tf < 0 ? -tf : tf;
                    tf = tf >= p.T_full ? 2 * (p.T_full - 1) - tf : tf;
                    a4[g] = bload(rsA, ((unsigned)(a_rb + (tf >> p.tshift) * p.V) * (unsigned)p.lda + (unsigned)cc) * 4u, 0u);
                }
                b4[g] = bload(rsW, w_off, (unsigned)kb * 4u);
            }
        }
        if (p.a_lrelu) {
#pragma unroll
            for (int g = 0; g < GMAX; ++g) {
                a4[g][0] = lrelu02(a4[g][0]); a4[g][1] = lrelu02(a4[g][1]);
                a4[g][2] = lrelu02(a4[g][2]); a4[g][3] = lrelu02(a4[g][3]);
            }
        }
#pragma unroll
        for (int g = 0; g < GMAX; ++g) {
            if (g0 + g < g_end) {                   // wave-uniform
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(b4[g][0], a4[g][0], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(b4[g][1], a4[g][1], acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(b4[g][2], a4[g][2], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(b4[g][3], a4[g][3], acc1, 0, 0, 0);
            }
        }
    }
    f32x4 acc = acc0 + acc1;
    if (wave > 0) red[wave - 1][lane] = acc;
    __syncthreads();
    if (wave > 0) return;
    acc = ((acc + red[0][lane]) + red[1][lane]) + red[2][lane];
    if (row >= p.M || col >= p.N) return;           // N % 4 == 0 is checked on the host for this kernel
    f32x4 v = acc + ebias;
    if (p.act == 1) { v[0] = gelu_erf(v[0]); v[1] = gelu_erf(v[1]); v[2] = gelu_erf(v[2]); v[3] = gelu_erf(v[3]); }
    else if (p.act == 2) { v[0] = lrelu02(v[0]); v[1] = lrelu02(v[1]); v[2] = lrelu02(v[2]); v[3] = lrelu02(v[3]); }
    else if (p.act == 3) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f); }
    v += eres;
    *reinterpret_cast<f32x4*>(p.C + (size_t)row * p.ldc + col) = v;
}

// the 16 x 16 variant serves up to 8 windows (and the style MLP's single rows); K in whole groups of 16
bool gemm_is_skinny16(const GemmParams& p) {
    // measured (characterize against a 585-row bank, ms per call): up to 2 windows 0.31 / 0.34 either way; 3 windows 0.445 -> 0.399,
    // 4: 0.458 -> 0.416, 6: 0.512 -> 0.498, 8: 0.598 -> 0.591 with the 16 x 16 tiles; beyond 768 rows no gain (the weights are
    // re-read once per 16 rows)
    // (and only while the weight matrix is small: it is re-read once per 16 rows - the decoder's block-diagonal style GEMM, 1024 x 1024,
    // took 44 us at 585 rows here against 2 x 15.6 us for the two per-layer launches on the 32 x 32 tiles)
    return gemm_is_skinny(p) && p.K % 16 == 0 && (p.M <= 192 || (p.M <= 768 && (long long)p.N * p.K <= 512 * 1024));
}

template <int BN>
static constexpr size_t lds_bytes() { return (size_t)(BM + BN) * LDSK * sizeof(float); }

hipError_t gemm_init() {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&mocha_gemm_f32<128, 2, 2, 2, 2>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes<128>());
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(&mocha_gemm_f32<64, 4, 1, 1, 2>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes<64>());
    if (e != hipSuccess) return e;
    return hipFuncSetAttribute(reinterpret_cast<const void*>(&mocha_gemm_f32<64, 2, 2, 1, 1>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)((64 + 64) * LDSK * sizeof(float)));
}

// Mid-size launches (a few dozen windows: too many rows for the skinny kernel, fewer 128 x 64 tiles than the chip has
// slots): 64 x 64 tiles, four waves of 32 x 32, so that twice as many workgroups share the CUs.
bool gemm_is_small(const GemmParams& p) {
    if (gemm_is_skinny(p) || p.ksplit > 1) return false;
    const long long t_narrow = (long long)((p.M + BM - 1) / BM) * ((p.N + 63) / 64);
    return t_narrow < 768;
}

bool gemm_is_narrow(const GemmParams& p) {
    // Tile choice.  Every tile costs the same matrix-pipe time, so a launch takes about ceil(tiles / 256 CUs) tile-times;
    // with few tiles per CU the rounding is expensive (824 tiles of 128x128 -> 3.2 per CU, a 4-tile critical path), and the
    // 128x64 tile has twice the tiles.  Since the operand fetches are buffer loads the two tiles run at the same rate per
    // tile-FLOP on this path (measured: 110 vs 108 TFLOP/s), so the narrow tile wins whenever its balance is not clearly
    // worse; the wide one is kept for launches where it balances > 10 % better.
    const int m_tiles = (p.M + BM - 1) / BM;
    auto balance = [](long long tiles) { return (tiles / 256.0) / (double)((tiles + 255) / 256); };
    const long long t_wide = (long long)m_tiles * ((p.N + 127) / 128) * p.ksplit;
    const long long t_narrow = (long long)m_tiles * ((p.N + 63) / 64) * p.ksplit;
    // long K loops (the matcher's 23 040) amortise prologue / epilogue and run 5 % faster on the wide tile (129 vs 123 TFLOP/s)
    const double th = p.K >= 4096 ? 1.05 : 0.90;
    return ((p.N % 128 != 0) && (p.N <= 256)) || balance(t_narrow) > th * balance(t_wide);
}

hipError_t launch_gemm(const GemmParams& p, hipStream_t s) {
    if (p.M <= 0 || p.N <= 0) return hipSuccess;
    if (p.K % BK != 0) return hipErrorInvalidValue;
    if (p.gather && (p.Cc % BK != 0)) return hipErrorInvalidValue;
    // 32-bit buffer offsets: a gathered source is addressed from its base, a tile of plain A / W from the tile's first row
    if (p.gather && (long long)p.M / p.T_out * p.T_src * p.lda * 4 >= (1ll << 31)) return hipErrorInvalidValue;
    if (128ll * p.lda * 4 >= (1ll << 31) || 128ll * p.K * 4 >= (1ll << 31)) return hipErrorInvalidValue;
    if (gemm_is_skinny16(p)) {
        dim3 grid(((p.M + 15) / 16) * ((p.N + 15) / 16));
        hipLaunchKernelGGL(mocha_gemm_skinny16, grid, dim3(256), 0, s, p);
        return hipGetLastError();
    }
    if (gemm_is_skinny(p)) {
        dim3 grid(((p.M + 31) / 32) * ((p.N + 31) / 32));
        hipLaunchKernelGGL(mocha_gemm_skinny, grid, dim3(256), 0, s, p);
        return hipGetLastError();
    }
    if (gemm_is_small(p)) {
        const int mt = (p.M + 63) / 64;
        const int mp = mt >= 8 ? (mt + 7) / 8 * 8 : mt;
        hipLaunchKernelGGL((mocha_gemm_f32<64, 2, 2, 1, 1>), dim3(mp * ((p.N + 63) / 64)), dim3(256), (64 + 64) * LDSK * sizeof(float), s, p);
        return hipGetLastError();
    }
    const int m_tiles = (p.M + BM - 1) / BM;
    const int m_pad = m_tiles >= 8 ? (m_tiles + 7) / 8 * 8 : m_tiles;
    if (gemm_is_narrow(p)) {
        dim3 grid(m_pad * ((p.N + 63) / 64), 1, p.ksplit);
        hipLaunchKernelGGL((mocha_gemm_f32<64, 4, 1, 1, 2>), grid, dim3(256), lds_bytes<64>(), s, p);
    } else {
        dim3 grid(m_pad * ((p.N + 127) / 128), 1, p.ksplit);
        hipLaunchKernelGGL((mocha_gemm_f32<128, 2, 2, 2, 2>), grid, dim3(256), lds_bytes<128>(), s, p);
    }
    return hipGetLastError();
}

}  // namespace mocha
