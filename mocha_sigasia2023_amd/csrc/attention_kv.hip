// Decoder cross-attention on pre-split key / value images (head dim 256, <= 96 tokens), bf16 matrix pipe, fp32-accurate.
// Reference: Attention.forward with adain=True, net/transformer.py:49-76.  Same arithmetic as attention_x3.hip (three-plane operands,
// six v_mfma_f32_32x32x16_bf16 passes per product, fp32 softmax in registers); what differs is where the operands come from.
//
// With the decoder fold (mocha_api.cpp) the keys IN(cha) and the values cha are the SAME for the four heads and for both
// decoder layers.  mocha_attention_x3<256> nevertheless fetched, plane-split and staged them once per (window, head, layer): eight
// times.  Here the instance norm that produces IN(cha) writes both operands ONCE, already split into bf16 planes and already in the
// order the LDS stage wants (mocha_instnorm, InormExtra::kvimg):
//
//   image of a window = 16 stages of 18 432 B:  stage c < 8  = K, head dims 32c .. 32c+31;  stage 8 + c = V, head dims 32c .. 32c+31
//   stage = [plane 3][row 96][64 B]: a row's 32 dims as bf16.  K rows are stored with their four 16-byte pieces XOR-swizzled by
//   (row >> 2) & 3 (ds_read_b128 of 16 rows x one piece then hits every bank once); V rows are plain (ds_read_b64_tr_b16 reads 4 rows x
//   64 B per half wave: every bank once).  Rows n .. 95 are zero.
//
// One workgroup of TWELVE waves per window: wave = (head, query block of 32) - three waves on every SIMD of the CU whatever order the
// dispatcher deals them in (a six-wave workgroup per head pair measured 1.2 resident waves per SIMD: the second workgroup of a CU needs
// 2 + 2 + 1 + 1 free wave slots in the right places, and at 164 VGPRs a SIMD has three).  A stage is copied global -> registers
// -> LDS with one or two 16-byte pieces per thread (no arithmetic), double-buffered, ONE barrier per stage; the twelve waves share it.  Queries
// never touch LDS: the B operand of the 32x32x16 MFMA is eight consecutive head dims of the lane's own query, i.e. two 16-byte global
// loads, split into planes in registers (each query element is split exactly once), fetched two chunks ahead.
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

static constexpr int KV_STG = ATTN_KV_STAGE_BYTES;      // 18 432
static constexpr int KV_PLANE = KV_STG / 3;             // 6 144

template <int DH, int HPW>
__global__ __launch_bounds__(192 * HPW) __attribute__((amdgpu_waves_per_eu(HPW == 2 ? 4 : 3, HPW == 2 ? 4 : 3))) void mocha_attention_x3_kv(AttnKvParams p) {
    // HPW == 2 (six waves per head pair) is built for 128 registers: two workgroups per CU then fit however the dispatcher deals their
    // waves over the SIMDs.  It keeps one query chunk in flight instead of two and reads one key tile's fragments at a time.
    constexpr bool LEAN = HPW == 2;
    constexpr int NKT = 3, NS = DH / 32;                 // key tiles; stages per operand
    constexpr int NTHR = 192 * HPW;
    constexpr int NPC = (1152 + NTHR - 1) / NTHR;        // 16-byte pieces of a stage per thread: 3 (six waves) or 2 (twelve: the second one for the first half)
    __shared__ __attribute__((aligned(16))) unsigned char sm[2 * KV_STG];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int hp = wave / 3, w = wave - 3 * hp;          // head of the pair, query block
    const int l31 = lane & 31, hh = lane >> 5;
    const int id = blockIdx.x;                           // XCD-aware order: the workgroups of a window are 8 ids apart (same XCD: the
    const int slot = id >> 3;                            // later ones find the image in that XCD's L2)
    const int ngrp = p.heads / HPW;
    const int grp = slot % ngrp;
    const int b = (slot / ngrp) * 8 + (id & 7);
    if (b >= p.B) return;
    const int head = grp * HPW + hp;
    const int nq = p.nq, nk = p.nk;
    const int query = w * 32 + l31;

    const __amdgpu_buffer_rsrc_t rsi = make_rsrc(reinterpret_cast<const unsigned char*>(p.kv) + (size_t)b * ATTN_KV_IMG_BYTES);
    const __amdgpu_buffer_rsrc_t rsq = make_rsrc(p.q + (size_t)b * nq * p.ldq + head * DH);
    const unsigned q_off = (unsigned)((query < nq ? query : nq - 1) * p.ldq + hh * 8) * 4u;

    // ---- stage copy: piece i = r * NTHR + tid of the stage's 1152
    u32x4 sr[NPC];
    auto fetch_stage = [&](int s) __attribute__((always_inline)) {
#pragma unroll
        for (int r = 0; r < NPC; ++r)
            if ((r + 1) * NTHR <= 1152 || r * NTHR + tid < 1152)
                sr[r] = __builtin_bit_cast(u32x4, bload(rsi, (unsigned)(r * NTHR + tid) * 16u, (unsigned)s * (unsigned)KV_STG));
    };
    auto write_stage = [&](int s) __attribute__((always_inline)) {
#pragma unroll
        for (int r = 0; r < NPC; ++r)
            if ((r + 1) * NTHR <= 1152 || r * NTHR + tid < 1152)
                *reinterpret_cast<u32x4*>(sm + (s & 1) * KV_STG + (r * NTHR + tid) * 16) = sr[r];
    };
    // ---- queries: chunk c, k step ks -> head dims 32c + 16ks + 8hh .. + 7 of this lane's query
    f32x4 qr[LEAN ? 1 : 2][4];
    auto fetch_q = [&](int c, auto set) __attribute__((always_inline)) {
        constexpr int S = decltype(set)::value;
#pragma unroll
        for (int i = 0; i < 4; ++i) qr[S][i] = bload(rsq, q_off + (unsigned)((i >> 1) * 64 + (i & 1) * 16), (unsigned)c * 128u);
    };
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, LEAN ? 0 : 1>;

    f32x16 st[NKT];
#pragma unroll
    for (int t = 0; t < NKT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) st[t][r] = 0.f;

    fetch_stage(0);
    fetch_q(0, S0{});
    if (!LEAN) fetch_q(1, S1{});
    write_stage(0);
    fetch_stage(1);

    // K fragment address inside a stage: row (t * 32 + l31) * 64 B, piece (2 ks + hh) ^ ((row >> 2) & 3)
    const int ksw = (l31 >> 2) & 3;
    // ---------------- phase 1: S^T[key][query] over eight K stages
    auto s_chunk = [&](int c, auto set) __attribute__((always_inline)) {
        constexpr int S = decltype(set)::value;
        __syncthreads();                                  // stage c is in LDS; the other buffer is free
        const unsigned char* buf = sm + (c & 1) * KV_STG;
        auto split_q = [&](int ks, s16x8 (&bq)[3]) __attribute__((always_inline)) {
            u32x2 lo[3], hi[3];
            plane_split4(qr[S][2 * ks], lo);
            plane_split4(qr[S][2 * ks + 1], hi);
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const u32x4 v = {lo[q][0], lo[q][1], hi[q][0], hi[q][1]};
                bq[q] = __builtin_bit_cast(s16x8, v);
            }
        };
        if (LEAN) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                s16x8 bq[3];
                split_q(ks, bq);
                if (ks == 1 && c + 1 < NS) fetch_q(c + 1, set);      // the chunk's registers are free
                const int pc = ((2 * ks + hh) ^ ksw) * 16;
                // one key tile at a time: three fragments live, six accumulating MFMAs on the tile's scores (the accumulator forwards)
#pragma unroll
                for (int t = 0; t < NKT; ++t) {
                    s16x8 a[3];
#pragma unroll
                    for (int q = 0; q < 3; ++q) a[q] = *reinterpret_cast<const s16x8*>(buf + q * KV_PLANE + (t * 32 + l31) * 64 + pc);
#pragma unroll
                    for (int pr = 0; pr < 6; ++pr)
                        st[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[PLANE_PA[pr]], bq[PLANE_PB[pr]], st[t], 0, 0, 0);
                }
            }
        } else {
            s16x8 bq[2][3];
            split_q(0, bq[0]);
            split_q(1, bq[1]);
            if (c + 2 < NS) fetch_q(c + 2, set);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const int pc = ((2 * ks + hh) ^ ksw) * 16;
                s16x8 a[3][NKT];
#pragma unroll
                for (int q = 0; q < 3; ++q)
#pragma unroll
                    for (int t = 0; t < NKT; ++t) a[q][t] = *reinterpret_cast<const s16x8*>(buf + q * KV_PLANE + (t * 32 + l31) * 64 + pc);
#pragma unroll
                for (int pr = 0; pr < 6; ++pr)
#pragma unroll
                    for (int t = 0; t < NKT; ++t)
                        st[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[PLANE_PA[pr]][t], bq[ks][PLANE_PB[pr]], st[t], 0, 0, 0);
            }
        }
        write_stage(c + 1);                               // stage c + 1 (K, or V pass 0 after the last K stage): its buffer has been free since the barrier
        fetch_stage(c + 2);                               // c + 2 <= 9 < 16
    };
    for (int c = 0; c < NS; c += 2) {
        s_chunk(c, S0{});
        s_chunk(c + 1, S1{});
    }

    // ---------------- phase 2: softmax over keys for this lane's query (fp32, as attention_x3.hip)
    // st[t][r] = S[query][key = 32t + (r&3) + 8(r>>2) + 4hh]
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

    // ---------------- phase 3: O^T[d][query] = sum_key V[key][d] P^T[key][query], 32 head dims per V stage
    float* og = p.out + ((size_t)b * nq + query) * p.ldo + head * DH;
    // transposed read (attention_x3.hip): lane L of a 16-lane group supplies the address of row (L & 15) >> 2, columns 4 (L & 3) ..; it receives
    // column L & 15 of the four rows.  Group g = lane >> 4: dims 16 (g & 1) .. + 15 of the 32-dim block, key half h = g >> 1.  Rows are 32 bf16.
    const int tr_base = ((4 * hh + ((lane & 15) >> 2)) * 32 + 16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2;      // bytes
#pragma unroll 1
    for (int dp = 0; dp < NS; ++dp) {
        const int s = NS + dp;
        __syncthreads();
        const unsigned char* buf = sm + (s & 1) * KV_STG;
        f32x16 o;
#pragma unroll
        for (int r = 0; r < 16; ++r) o[r] = 0.f;
#pragma unroll
        for (int t = 0; t < NKT; ++t)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                s16x8 va[3];
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    const unsigned char* src = buf + q * KV_PLANE + tr_base + (32 * t + 16 * j) * 64;
                    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(src));
                    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(src + 8 * 64));
                    va[q] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                }
#pragma unroll
                for (int pr = 0; pr < 6; ++pr)
                    o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va[PLANE_PA[pr]], pp[t][j][PLANE_PB[pr]], o, 0, 0, 0);
            }
        if (s + 1 < 2 * NS) write_stage(s + 1);
        if (s + 2 < 2 * NS) fetch_stage(s + 2);
        // o[r] = O[query][dcol = dp*32 + (r&3) + 8(r>>2) + 4hh]: regs 4g..4g+3 are 4 consecutive dims
        if (query < nq) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 v = {o[4 * g], o[4 * g + 1], o[4 * g + 2], o[4 * g + 3]};
                *reinterpret_cast<f32x4*>(og + dp * 32 + 8 * g + 4 * hh) = v;
            }
        }
    }
}

hipError_t launch_attention_x3_kv(const AttnKvParams& p, hipStream_t s) {
    if (p.B <= 0) return hipSuccess;
    if (p.nq < 1 || p.nk < 1 || p.nq > 96 || p.nk > 96 || p.dh != 256 || (p.heads & 1) || p.heads < 2) return hipErrorInvalidValue;
    if (p.heads % 4 == 0 && !p.pairs) {
        const unsigned grid = (unsigned)(((p.B + 7) / 8) * 8 * (p.heads / 4));
        hipLaunchKernelGGL((mocha_attention_x3_kv<256, 4>), dim3(grid), dim3(768), 0, s, p);
    } else {
        const unsigned grid = (unsigned)(((p.B + 7) / 8) * 8 * (p.heads / 2));
        hipLaunchKernelGGL((mocha_attention_x3_kv<256, 2>), dim3(grid), dim3(384), 0, s, p);
    }
    return hipGetLastError();
}

}  // namespace mocha
