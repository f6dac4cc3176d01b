// Multi-head softmax attention over <= 96 tokens on the bf16 matrix pipe of gfx950, fp32-accurate
// (reference: Attention.forward, net/transformer.py:65-76; same interface and results as attention.hip).
//
// Both contractions run as plane products (gemm_x3.hip): every fp32 operand is the exact sum of three bf16 values, six
// v_mfma_f32_32x32x16_bf16 passes (fp32 accumulate) reproduce the fp32 product to within one fp32 rounding - at 6/16 of the fp32 MFMA's cycles.
//
// One workgroup of three waves per (window, head); wave w owns query block 32w..32w+31, keys are padded to 96.
//   1. S^T = K · Q^T, head dim in chunks of 32: a thread fetches four 16-byte pieces of K and of Q one chunk ahead, splits them
//      into planes and stores them as [plane][k step][k half][row][8 bf16] (the x3 GEMM's image: contiguous ds_read_b128 fragments,
//      conflict-free b64 stores).  Keys on the MFMA rows, queries on the lanes: the softmax over keys is a per-lane reduction.
//   2. softmax in registers, fp32 (as attention.hip).
//   3. O^T = V^T · P^T.  The accumulator tile P^T (keys in registers, query on the lane) is split into three planes in registers:
//      registers 8j..8j+7 of a tile are, for lane half h, the keys 16j + 8(e>>2) + 4h + (e&3), e = 0..7 - one B operand of a
//      K = 16 MFMA with that key order.  V is staged 64 head dims at a time as row-major planes [key][64 dims] and read through
//      ds_read_b64_tr_b16, which hands lane (dim, h) the four consecutive keys of its column: two reads give the A operand in the
//      same key order.  No transposing stores, P never leaves the register file.
#include "kernels.h"
#include <type_traits>
#include "device_utils.h"

namespace mocha {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

static constexpr int AX_ROWS = 96;                       // keys / queries, padded
static constexpr int AX_BLK = AX_ROWS * 8 + 16;          // bf16 per (k step, k half) block of a plane: 1536 B + 32 B (b64 stores cover all banks)
static constexpr int AX_PLANE = 4 * AX_BLK;              // a 32-wide chunk: 2 k steps x 2 k halves
static constexpr int AX_OPER = 3 * AX_PLANE;             // K or Q chunk, three planes: 9408 bf16
static constexpr int AX_VPLANE = AX_ROWS * 64;           // V pass plane: [key][64 dims]
static constexpr int AX_LDS = 2 * AX_OPER > 3 * AX_VPLANE ? 2 * AX_OPER : 3 * AX_VPLANE;     // 18 816 bf16 = 37 632 B

// PF2 (a handful of windows: fewer workgroups than CUs, so the kernel is a chain of memory round trips - one per 32-dim K/Q chunk and
// per 64-dim V pass): chunks are fetched TWO steps ahead into alternating register sets, and the first two V passes are already in
// flight while the softmax runs.  Same arithmetic in the same order: results are bit-identical to the one-step-ahead variant, which
// keeps its 3 waves per SIMD for full batches.
template <int DH, bool PF2>
__global__ __launch_bounds__(192) __attribute__((amdgpu_waves_per_eu(PF2 ? 2 : 3, PF2 ? 2 : 3))) void mocha_attention_x3(AttnParams p) {
    constexpr int NTHR = 192, NKT = 3;
    __shared__ __attribute__((aligned(16))) unsigned short sm[AX_LDS];
    unsigned short* Ks = sm;
    unsigned short* Qs = sm + AX_OPER;
    unsigned short* Vs = sm;                             // the V passes reuse the chunk stages

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hh = lane >> 5;
    const int id = blockIdx.x;                           // XCD-aware order as in attention.hip: the heads of a window share an XCD
    const int slot = id >> 3;
    const int head = slot % p.heads;
    const int b = (slot / p.heads) * 8 + (id & 7);
    if (b >= p.B) return;                                // whole workgroup: EXEC stays all ones for the transposed reads below
    const int nq = p.nq, nk = p.nk;

    const float* qg = p.q + (size_t)b * nq * p.ldq + head * DH;
    long long bk = b;                                         // keys / values of window b: its own rows, or the bank entry kv_idx[b] (clamped)
    if (p.kv_idx) { bk = p.kv_idx[b]; bk = bk < 0 ? 0 : (bk >= p.kv_rows ? p.kv_rows - 1 : bk); }
    const float* kg = p.k + (size_t)bk * nk * p.ldk + head * (p.hsk < 0 ? DH : p.hsk);
    const float* vg = p.v + (size_t)bk * nk * p.ldv + head * (p.hsv < 0 ? DH : p.hsv);
    const __amdgpu_buffer_rsrc_t rsq = make_rsrc(qg), rsk = make_rsrc(kg), rsv = make_rsrc(vg);

    // ---------------- phase 1: S^T[key][query], head dim in chunks of 32
    f32x16 st[NKT];
#pragma unroll
    for (int t = 0; t < NKT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) st[t][r] = 0.f;

    // a chunk of K (96 x 32) and of Q (96 x 32): 768 16-byte pieces each, four per thread; rows beyond nk / nq are clamped to the last
    // valid row (the scores of padded keys are masked below, padded queries are never stored)
    f32x4 kr[PF2 ? 2 : 1][4], qr[PF2 ? 2 : 1][4];
    unsigned k_off[4], q_off[4];
    int st_off[4];                                       // bf16 offset of this piece inside a plane of the chunk image
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int f = tid + NTHR * i;
        const int row = f >> 3, c = f & 7;               // piece c of the row's eight: k step c >> 2, k half (c >> 1) & 1, 8-byte half c & 1
        const int rk = row < nk ? row : nk - 1, rq = row < nq ? row : nq - 1;
        k_off[i] = (unsigned)(rk * p.ldk + c * 4) * 4u;
        q_off[i] = (unsigned)(rq * p.ldq + c * 4) * 4u;
        st_off[i] = (c >> 1) * AX_BLK + row * 8 + (c & 1) * 4;
    }
    auto fetch_kq = [&](int c, auto set) __attribute__((always_inline)) {
        constexpr int S = decltype(set)::value;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            kr[S][i] = bload(rsk, k_off[i], (unsigned)c * 128u);
            qr[S][i] = bload(rsq, q_off[i], (unsigned)c * 128u);
        }
    };
    auto stage_kq = [&](auto set) __attribute__((always_inline)) {
        constexpr int S = decltype(set)::value;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            u32x2 pk[3], pq[3];
            plane_split4(kr[S][i], pk);
            plane_split4(qr[S][i], pq);
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                *reinterpret_cast<u32x2*>(Ks + q * AX_PLANE + st_off[i]) = pk[q];
                *reinterpret_cast<u32x2*>(Qs + q * AX_PLANE + st_off[i]) = pq[q];
            }
        }
    };
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, PF2 ? 1 : 0>;
    constexpr int NC = DH / 32, NP = DH / 64;
    fetch_kq(0, S0{});
    if (PF2) fetch_kq(1, S1{});
    stage_kq(S0{});
    __syncthreads();
    auto s_chunk = [&](int c, auto cur_set, auto nxt_set) __attribute__((always_inline)) {
        // PF2: the set chunk c was staged from is free - chunk c + 2 goes there; otherwise chunk c + 1 into the only set
        if (PF2) { if (c + 2 < NC) fetch_kq(c + 2, cur_set); }
        else if (c + 1 < NC) fetch_kq(c + 1, S0{});
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            s16x8 a[3][NKT], bq[3];
            const int blk = (ks * 2 + hh) * AX_BLK;
#pragma unroll
            for (int q = 0; q < 3; ++q) {
#pragma unroll
                for (int t = 0; t < NKT; ++t) a[q][t] = *reinterpret_cast<const s16x8*>(Ks + q * AX_PLANE + blk + (t * 32 + l31) * 8);
                bq[q] = *reinterpret_cast<const s16x8*>(Qs + q * AX_PLANE + blk + (wave * 32 + l31) * 8);
            }
#pragma unroll
            for (int pr = 0; pr < 6; ++pr)
#pragma unroll
                for (int t = 0; t < NKT; ++t)
                    st[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[PLANE_PA[pr]][t], bq[PLANE_PB[pr]], st[t], 0, 0, 0);
        }
        __syncthreads();
        if (c + 1 < NC) stage_kq(nxt_set);
        __syncthreads();
    };
    for (int c = 0; c < NC; c += 2) {                   // NC is even: the register sets alternate statically
        s_chunk(c, S0{}, S1{});
        s_chunk(c + 1, S1{}, S0{});
    }

    // a V pass (96 keys x 64 dims): 1536 pieces, eight per thread; 16 lanes cover a key's 256 bytes
    f32x4 vr[PF2 ? 2 : 1][8];
    unsigned v_off[8];
    int vs_off[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int f = tid + NTHR * i;
        const int row = f >> 4, c4 = (f & 15) * 4;
        const int rv = row < nk ? row : nk - 1;          // padded keys carry P = 0
        v_off[i] = (unsigned)(rv * p.ldv + c4) * 4u;
        vs_off[i] = row * 64 + c4;
    }
    auto fetch_v = [&](int dp, auto set) __attribute__((always_inline)) {
        constexpr int S = decltype(set)::value;
#pragma unroll
        for (int i = 0; i < 8; ++i) vr[S][i] = bload(rsv, v_off[i], (unsigned)dp * 256u);
    };
    auto stage_v = [&](auto set) __attribute__((always_inline)) {
        constexpr int S = decltype(set)::value;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            u32x2 pv[3];
            plane_split4(vr[S][i], pv);
#pragma unroll
            for (int q = 0; q < 3; ++q) *reinterpret_cast<u32x2*>(Vs + q * AX_VPLANE + vs_off[i]) = pv[q];
        }
    };
    if (PF2) { fetch_v(0, S0{}); if (NP > 1) fetch_v(1, S1{}); }      // in flight under the softmax

    // ---------------- phase 2: softmax over keys for this lane's query (fp32, as attention.hip)
    // st[t][r] = S[query = 32*wave + l31][key = 32t + (r&3) + 8(r>>2) + 4hh]
    float mx = -INFINITY;
#pragma unroll
    for (int t = 0; t < NKT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = t * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
            if (key >= nk) st[t][r] = -INFINITY;
            mx = fmaxf(mx, st[t][r]);
        }
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    const float c2 = p.scale * 1.44269504088896340736f;
    const float mb = -mx * c2;
    float sum = 0.f;
#pragma unroll
    for (int t = 0; t < NKT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float e = __builtin_amdgcn_exp2f(fmaf(st[t][r], c2, mb));
            st[t][r] = e;
            sum += e;
        }
    sum += __shfl_xor(sum, 32);
    const float inv = 1.0f / sum;

    // P^T planes: tile t, k block j -> registers 8j..8j+7 of st[t], normalised, as three bf16 planes (B operands)
    s16x8 pp[NKT][2][3];
#pragma unroll
    for (int t = 0; t < NKT; ++t)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const f32x4 lo = {st[t][8 * j] * inv, st[t][8 * j + 1] * inv, st[t][8 * j + 2] * inv, st[t][8 * j + 3] * inv};
            const f32x4 hi = {st[t][8 * j + 4] * inv, st[t][8 * j + 5] * inv, st[t][8 * j + 6] * inv, st[t][8 * j + 7] * inv};
            u32x2 a[3], b2[3];
            plane_split4(lo, a); plane_split4(hi, b2);
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const u32x4 v = {a[q][0], a[q][1], b2[q][0], b2[q][1]};
                pp[t][j][q] = __builtin_bit_cast(s16x8, v);
            }
        }

    // ---------------- phase 3: O^T[d][query] = sum_key V[key][d] * P^T[key][query], 64 head dims per pass
    const int query = wave * 32 + l31;
    float* og = p.out + ((size_t)b * nq + query) * p.ldo + head * DH;
    // transposed read: lane L of a 16-lane group supplies the address of row (L & 15) >> 2, columns 4 (L & 3) ..; it receives
    // column L & 15 of the four rows.  Group g = lane >> 4: dims 16 (g & 1) .. + 15 of the 32-dim block, key half h = g >> 1.
    const int tr_base = ((4 * hh + ((lane & 15) >> 2)) * 64 + 16 * ((lane >> 4) & 1) + 4 * (lane & 3));
    if (!PF2) fetch_v(0, S0{});                          // PF2: passes 0 and 1 were fetched before the softmax (below)
    stage_v(S0{});                                       // the chunk stages are free: every wave passed the last barrier of phase 1
    __syncthreads();
    auto v_pass = [&](int dp, auto cur_set, auto nxt_set) __attribute__((always_inline)) {
        if (PF2) { if (dp + 2 < NP) fetch_v(dp + 2, cur_set); }
        else if (dp + 1 < NP) fetch_v(dp + 1, S0{});
        f32x16 o[2];
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[d][r] = 0.f;
#pragma unroll
        for (int t = 0; t < NKT; ++t)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int d = 0; d < 2; ++d) {
                    s16x8 va[3];
#pragma unroll
                    for (int q = 0; q < 3; ++q) {
                        const unsigned short* src = Vs + q * AX_VPLANE + tr_base + (32 * t + 16 * j) * 64 + 32 * d;
                        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(src));
                        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(src + 8 * 64));
                        va[q] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                    }
#pragma unroll
                    for (int pr = 0; pr < 6; ++pr)
                        o[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va[PLANE_PA[pr]], pp[t][j][PLANE_PB[pr]], o[d], 0, 0, 0);
                }
        // o[d][r] = O[query][dcol = dp*64 + d*32 + (r&3) + 8(r>>2) + 4hh]: regs 4g..4g+3 are 4 consecutive dims
        if (query < nq) {
#pragma unroll
            for (int d = 0; d < 2; ++d)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    f32x4 w = {o[d][4 * g], o[d][4 * g + 1], o[d][4 * g + 2], o[d][4 * g + 3]};
                    *reinterpret_cast<f32x4*>(og + dp * 64 + d * 32 + 8 * g + 4 * hh) = w;
                }
        }
        __syncthreads();
        if (dp + 1 < NP) stage_v(nxt_set);
        __syncthreads();
    };
    for (int dp = 0; dp < NP; dp += 2) {                 // NP is even
        v_pass(dp, S0{}, S1{});
        v_pass(dp + 1, S1{}, S0{});
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// A handful of windows (streaming: ONE window, two or four (window, head) pairs on a 256-CU chip): the launch above is then a serial
// chain per wave - eight K/Q chunks and four V passes at head dim 256, each a staging step, two barriers and 36 / 108 MFMAs - and
// that chain, not the memory system, is what the caller waits for.  This variant gives a (window, head) pair TWELVE waves, four
// groups of three: group g contracts a quarter of the head dim for all 96 x 96 scores (its own K/Q stages), the four partial score
// tiles are summed through LDS in a fixed order by every group (so all groups hold bit-identical scores and repeat the cheap
// softmax), and group g then produces a quarter of the output columns (one 64-dim V pass at head dim 256; half of one at 128).
// Per wave: 2 chunks + 1 pass instead of 8 + 4.  The score sums are associated differently from the kernel above (four partial sums
// instead of one running sum), so results agree to fp32 rounding, not bit for bit.
template <int DH>
__global__ __launch_bounds__(768) void mocha_attention_x3_split(AttnParams p) {
    constexpr int NTHR = 192, NKT = 3, NG = 4;
    constexpr int NC = DH / 32, NCG = NC / NG;           // K/Q chunks of 32 dims: in all, per group
    constexpr int ND = DH == 256 ? 2 : 1;                // 32-dim output blocks per group in phase 3
    extern __shared__ __attribute__((aligned(16))) unsigned short smx[];          // [NG][AX_LDS]
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int grp = wave / 3, w = wave - grp * 3;        // group, query block
    const int gtid = tid - grp * NTHR;
    const int l31 = lane & 31, hh = lane >> 5;
    unsigned short* Ks = smx + grp * AX_LDS;
    unsigned short* Qs = Ks + AX_OPER;
    unsigned short* Vs = Ks;

    const int id = blockIdx.x;
    const int slot = id >> 3;
    const int head = slot % p.heads;
    const int b = (slot / p.heads) * 8 + (id & 7);
    if (b >= p.B) return;
    const int nq = p.nq, nk = p.nk;
    const float* qg = p.q + (size_t)b * nq * p.ldq + head * DH;
    long long bk = b;                                         // keys / values of window b: its own rows, or the bank entry kv_idx[b] (clamped)
    if (p.kv_idx) { bk = p.kv_idx[b]; bk = bk < 0 ? 0 : (bk >= p.kv_rows ? p.kv_rows - 1 : bk); }
    const float* kg = p.k + (size_t)bk * nk * p.ldk + head * (p.hsk < 0 ? DH : p.hsk);
    const float* vg = p.v + (size_t)bk * nk * p.ldv + head * (p.hsv < 0 ? DH : p.hsv);
    const __amdgpu_buffer_rsrc_t rsq = make_rsrc(qg), rsk = make_rsrc(kg), rsv = make_rsrc(vg);

    // ---------------- phase 1: this group's quarter of S^T[key][query]
    f32x16 st[NKT];
#pragma unroll
    for (int t = 0; t < NKT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) st[t][r] = 0.f;
    f32x4 kr[NCG][4], qr[NCG][4];
    int st_off[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int f = gtid + NTHR * i;
        const int row = f >> 3, c = f & 7;
        const int rk = row < nk ? row : nk - 1, rq = row < nq ? row : nq - 1;
        const unsigned k_off = (unsigned)(rk * p.ldk + c * 4) * 4u, q_off = (unsigned)(rq * p.ldq + c * 4) * 4u;
        st_off[i] = (c >> 1) * AX_BLK + row * 8 + (c & 1) * 4;
#pragma unroll
        for (int j = 0; j < NCG; ++j) {                  // every chunk of the group is in flight from the start
            kr[j][i] = bload(rsk, k_off, (unsigned)(grp * NCG + j) * 128u);
            qr[j][i] = bload(rsq, q_off, (unsigned)(grp * NCG + j) * 128u);
        }
    }
    // this group's V columns (phase 3): pass dp of 64 dims
    const int dp = DH == 256 ? grp : grp >> 1;
    const int d0 = DH == 256 ? 0 : (grp & 1);
    f32x4 vr[8];
    int vs_off[8];
    auto fetch_v = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int f = gtid + NTHR * i;
            const int row = f >> 4, c4 = (f & 15) * 4;
            const int rv = row < nk ? row : nk - 1;
            vs_off[i] = row * 64 + c4;
            vr[i] = bload(rsv, (unsigned)(rv * p.ldv + c4) * 4u, (unsigned)dp * 256u);
        }
    };
#pragma unroll
    for (int j = 0; j < NCG; ++j) {
        if (j) __syncthreads();                          // every wave is done reading the previous chunk's stage
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            u32x2 pk[3], pq[3];
            plane_split4(kr[j][i], pk);
            plane_split4(qr[j][i], pq);
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                *reinterpret_cast<u32x2*>(Ks + q * AX_PLANE + st_off[i]) = pk[q];
                *reinterpret_cast<u32x2*>(Qs + q * AX_PLANE + st_off[i]) = pq[q];
            }
        }
        if (j == NCG - 1) fetch_v();                     // the chunk registers are free: V arrives under the MFMAs, the sum and the softmax
        __syncthreads();
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            s16x8 a[3][NKT], bq[3];
            const int blk = (ks * 2 + hh) * AX_BLK;
#pragma unroll
            for (int q = 0; q < 3; ++q) {
#pragma unroll
                for (int t = 0; t < NKT; ++t) a[q][t] = *reinterpret_cast<const s16x8*>(Ks + q * AX_PLANE + blk + (t * 32 + l31) * 8);
                bq[q] = *reinterpret_cast<const s16x8*>(Qs + q * AX_PLANE + blk + (w * 32 + l31) * 8);
            }
#pragma unroll
            for (int pr = 0; pr < 6; ++pr)
#pragma unroll
                for (int t = 0; t < NKT; ++t)
                    st[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[PLANE_PA[pr]][t], bq[PLANE_PB[pr]], st[t], 0, 0, 0);
        }
    }
    // ---------------- the four partial tiles of a query block, summed in group order by every group
    __syncthreads();
    {
        float* red = reinterpret_cast<float*>(Ks);       // [query block][tile quad 0..11][lane][4]
#pragma unroll
        for (int t = 0; t < NKT; ++t)
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) {
                const f32x4 v = {st[t][4 * qd], st[t][4 * qd + 1], st[t][4 * qd + 2], st[t][4 * qd + 3]};
                *reinterpret_cast<f32x4*>(red + ((w * 12 + t * 4 + qd) * 64 + lane) * 4) = v;
            }
    }
    __syncthreads();
#pragma unroll
    for (int t = 0; t < NKT; ++t)
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
            f32x4 acc = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(smx) + ((w * 12 + t * 4 + qd) * 64 + lane) * 4);
#pragma unroll
            for (int g = 1; g < NG; ++g) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(smx + g * AX_LDS) + ((w * 12 + t * 4 + qd) * 64 + lane) * 4);
                acc[0] += v[0]; acc[1] += v[1]; acc[2] += v[2]; acc[3] += v[3];
            }
            st[t][4 * qd] = acc[0]; st[t][4 * qd + 1] = acc[1]; st[t][4 * qd + 2] = acc[2]; st[t][4 * qd + 3] = acc[3];
        }

    // ---------------- phase 2: softmax over keys for this lane's query (as above)
    float mx = -INFINITY;
#pragma unroll
    for (int t = 0; t < NKT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = t * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
            if (key >= nk) st[t][r] = -INFINITY;
            mx = fmaxf(mx, st[t][r]);
        }
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    const float c2 = p.scale * 1.44269504088896340736f;
    const float mb = -mx * c2;
    float sum = 0.f;
#pragma unroll
    for (int t = 0; t < NKT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float e = __builtin_amdgcn_exp2f(fmaf(st[t][r], c2, mb));
            st[t][r] = e;
            sum += e;
        }
    sum += __shfl_xor(sum, 32);
    const float inv = 1.0f / sum;
    s16x8 pp[NKT][2][3];
#pragma unroll
    for (int t = 0; t < NKT; ++t)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const f32x4 lo = {st[t][8 * j] * inv, st[t][8 * j + 1] * inv, st[t][8 * j + 2] * inv, st[t][8 * j + 3] * inv};
            const f32x4 hi = {st[t][8 * j + 4] * inv, st[t][8 * j + 5] * inv, st[t][8 * j + 6] * inv, st[t][8 * j + 7] * inv};
            u32x2 a[3], b2[3];
            plane_split4(lo, a); plane_split4(hi, b2);
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const u32x4 v = {a[q][0], a[q][1], b2[q][0], b2[q][1]};
                pp[t][j][q] = __builtin_bit_cast(s16x8, v);
            }
        }

    // ---------------- phase 3: this group's output columns
    __syncthreads();                                     // every wave has read the partial sums: the stage becomes V's
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        u32x2 pv[3];
        plane_split4(vr[i], pv);
#pragma unroll
        for (int q = 0; q < 3; ++q) *reinterpret_cast<u32x2*>(Vs + q * AX_VPLANE + vs_off[i]) = pv[q];
    }
    __syncthreads();
    const int query = w * 32 + l31;
    float* og = p.out + ((size_t)b * nq + query) * p.ldo + head * DH + dp * 64 + d0 * 32;
    const int tr_base = ((4 * hh + ((lane & 15) >> 2)) * 64 + 16 * ((lane >> 4) & 1) + 4 * (lane & 3)) + 32 * d0;
    f32x16 o[ND];
#pragma unroll
    for (int d = 0; d < ND; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[d][r] = 0.f;
#pragma unroll
    for (int t = 0; t < NKT; ++t)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int d = 0; d < ND; ++d) {
                s16x8 va[3];
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    const unsigned short* src = Vs + q * AX_VPLANE + tr_base + (32 * t + 16 * j) * 64 + 32 * d;
                    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(src));
                    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(src + 8 * 64));
                    va[q] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                }
#pragma unroll
                for (int pr = 0; pr < 6; ++pr)
                    o[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va[PLANE_PA[pr]], pp[t][j][PLANE_PB[pr]], o[d], 0, 0, 0);
            }
    if (query < nq) {
#pragma unroll
        for (int d = 0; d < ND; ++d)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 v = {o[d][4 * g], o[d][4 * g + 1], o[d][4 * g + 2], o[d][4 * g + 3]};
                *reinterpret_cast<f32x4*>(og + d * 32 + 8 * g + 4 * hh) = v;
            }
    }
}

static constexpr size_t AX_SPLIT_LDS = (size_t)4 * AX_LDS * sizeof(unsigned short);     // 150 528 B
hipError_t attention_x3_init() {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(&mocha_attention_x3_split<256>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)AX_SPLIT_LDS);
}
// AttnParams::split_max (default 192): (window, head) pairs up to which the twelve-wave variant is launched (head dim 256: the
// decoder's four heads, 48 windows; whole decoder 109 vs 116 us at one window, 236 vs 244 at 32, equal at 64 windows, slower
// from 96 - tools/ab/attn_split_ab.py)

hipError_t launch_attention_x3(const AttnParams& p, hipStream_t s) {
    if (p.B <= 0) return hipSuccess;
    if (p.nq < 1 || p.nk < 1 || p.nq > 96 || p.nk > 96 || (p.dh != 128 && p.dh != 256)) return hipErrorInvalidValue;
    dim3 grid((unsigned)(((p.B + 7) / 8) * 8 * p.heads));
    const bool pf2 = (long long)p.B * p.heads <= 256;    // fewer (window, head) workgroups than CUs: latency-bound, prefetch two steps ahead
    if (p.dh == 256 && (long long)p.B * p.heads <= p.split_max) {
        // head dim 128 (the encoder) has one chunk per group left to split: measured equal to the two-steps-ahead variant, which it keeps
        hipLaunchKernelGGL((mocha_attention_x3_split<256>), grid, dim3(768), AX_SPLIT_LDS, s, p);
        return hipGetLastError();
    }
    if (p.dh == 128) {
        if (pf2) hipLaunchKernelGGL((mocha_attention_x3<128, true>), grid, dim3(192), 0, s, p);
        else hipLaunchKernelGGL((mocha_attention_x3<128, false>), grid, dim3(192), 0, s, p);
    } else {
        if (pf2) hipLaunchKernelGGL((mocha_attention_x3<256, true>), grid, dim3(192), 0, s, p);
        else hipLaunchKernelGGL((mocha_attention_x3<256, false>), grid, dim3(192), 0, s, p);
    }
    return hipGetLastError();
}

}  // namespace mocha
