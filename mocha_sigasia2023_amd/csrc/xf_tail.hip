// Fused transformer-layer tail for gfx950 (net/transformer.py:91-94, 23-34):
//     x1  = ao · Wo^T + bo + resid            attention output projection + residual
//     out = GELU(x1 · W1^T + b1) · W2^T + b2 + x1          feed-forward (exact-erf GELU) + residual
// in ONE launch: x1 and the 512-wide hidden activation never leave the register file.
//
// Why this shape.  As three GEMM launches the tail is 37 % of the demo step's GEMM time at K = 256 / 512 - short K loops
// where a quarter of a tile's life is prologue / epilogue - and round-trips a (rows x 512) hidden tensor through HBM.
// Here a WAVE owns 16 rows for the whole tail.  All contractions run on v_mfma_f32_16x16x4_f32 (exact f32, the same
// k-ordered fmaf chain as the 32x32x2 form) with the WEIGHT tile as the MFMA's "A" operand, so an accumulator holds the
// transposed result: lane & 15 = row, lane >> 4 = kg, registers 0..3 = columns 4 kg .. 4 kg + 3 of a 16-column tile.  That is
// exactly the "B" operand layout of the next contraction (B[k = kg][j = row], one k per register), provided the weight
// fragments use the k order k = 4 kg + r for step r - which is what one 16-byte LDS read per lane delivers.  So
//     x1 accumulators  ->  B operand of FF1,     GELU(FF1 accumulators)  ->  B operand of FF2
// with no lane movement and no LDS traffic for activations.  LDS only holds weight slabs (32 KB each, two slots per
// workgroup), filled by LDS-DMA (buffer_load ... lds) one slab ahead and shared by the 4 waves (64 rows) of a workgroup;
// two workgroups per CU (8 waves, <= 256 VGPRs each).  The hidden dimension is walked in 16 chunks of 32: FF1 chunk
// (K = 256) -> GELU -> its rank-32 update of the 16 x 256 output accumulators, so only 8 hidden registers are live.
// 64-row workgroups keep the tile quantisation at 6.4 -> 7 passes per CU on the demo step (92 %); the weight traffic from
// L2 is 24 KB per row-block and MFMA, 19 GB/s per CU.
//
// LDS slab layouts (fp32, unpadded; an LDS-DMA instruction writes 1 KB in lane order, the bank swizzle is applied by
// choosing which global 16-byte unit a lane fetches):
//   kind A  [256 rows n][32 k]   (Wo and W2 slabs): unit u of row r lives in slot u ^ ((r >> 1) & 7)
//   kind B  [32 rows n2][256 k]  (W1 slabs):        unit u of row r lives in slot u ^ (r & 15)
// both make the fragment reads (ds_read_b128: 16 rows x {kg, kg + 1} per service group) conflict-free.
#include "kernels.h"
#include "device_utils.h"

namespace mocha {

typedef float f32x4 __attribute__((ext_vector_type(4)));

#ifdef XT_EXP_NOGELU
__device__ __forceinline__ float xt_gelu(float x) { return x; }
#else
__device__ __forceinline__ float xt_gelu(float x) { return 0.5f * x * (1.0f + mocha_erf(x * 0.70710678118654752440f)); }
#endif

// Scheduling directive for one slab's compute block: 16 groups of 8 MFMAs; the two weight fragments of group g + 1 are read
// from LDS while group g multiplies (the scheduler otherwise issues read, wait, multiply in turn and exposes the LDS latency
// every 8 MFMAs), and VALU instructions (the GELU of the previous chunk) are spread VALU_PER between consecutive MFMAs.
template <int VALU_PER>
__device__ __forceinline__ void xt_pipeline() {
    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);                   // DS read x2: group 0's fragments
#pragma unroll
    for (int g = 0; g < 16; ++g) {
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);               // next group's fragments
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);           // MFMA
            if (VALU_PER) __builtin_amdgcn_sched_group_barrier(0x002, VALU_PER, 0);
        }
    }
}

static constexpr int XT_ROWS = 64;                  // rows per workgroup (4 waves x 16)
static constexpr int XT_SLAB = 8192;                // floats per slab (32 KB)

__global__ __launch_bounds__(256, 2) void mocha_xf_tail(XfTailParams p) {
    extern __shared__ __attribute__((aligned(16))) float xt_sm[];      // [2][XT_SLAB]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, kg = lane >> 4;
    const int row0 = blockIdx.x * XT_ROWS + wave * 16;
    int m = row0 + li;
    const bool mvalid = m < p.M;
    m = mvalid ? m : p.M - 1;

    // Slab sequence g = 0 .. S1 + 31:  Wo(0) .. Wo(S1-1), W1(0), then W1(c+1), W2(c) for c = 0 .. 14, W2(15): the FF1 of chunk
    // c + 1 is issued before the FF2 of chunk c, so that GELU(c) - straight-line VALU - sits in the same basic block as FF1(c+1)'s
    // MFMAs and runs in their shadow.
    const int S1 = p.Kin / 32;
    const int S = S1 + 32;
    const __amdgpu_buffer_rsrc_t rsWo = make_rsrc(p.Wo), rsW1 = make_rsrc(p.W1), rsW2 = make_rsrc(p.W2);
    const __amdgpu_buffer_rsrc_t rsA = make_rsrc(p.ao + (size_t)blockIdx.x * XT_ROWS * p.Kin);

    // ---- slab s -> LDS slot s & 1 (every wave issues 8 of the 32 one-KB pieces)
    const unsigned uA = (unsigned)((lane & 7) ^ ((4 * wave + (lane >> 4)) & 7));          // kind A: global unit for this lane's slot
    auto issue = [&](int s) __attribute__((always_inline)) {
        float* slot = xt_sm + (s & 1) * XT_SLAB;
        const int g = s - S1;                                           // position in the FF part: 0 -> W1(0); odd -> W1((g+1)/2); even -> W2(g/2-1); 31 -> W2(15)
        const bool w1 = g == 0 || (g > 0 && (g & 1) && g != 31);
        const unsigned cw = g <= 0 ? 0u : (g == 31 ? 15u : ((g & 1) ? (unsigned)(g + 1) >> 1 : ((unsigned)g >> 1) - 1u));
        if (!w1) {                                                      // kind A: [256 rows][32 k] of Wo (ld Kin) or W2 (ld 512)
            const bool wo = s < S1;
            const unsigned ld = wo ? (unsigned)p.Kin : 512u;
            const unsigned k0 = wo ? (unsigned)s * 32u : cw * 32u;
            const unsigned vo = ((unsigned)(8 * wave + (lane >> 3)) * ld + 4u * uA) * 4u;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (wo) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsWo, (__attribute__((address_space(3))) void*)(slot + (wave + 4 * i) * 256), 16, vo + (unsigned)i * 32u * ld * 4u, k0 * 4u, 0, 0);
                else __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW2, (__attribute__((address_space(3))) void*)(slot + (wave + 4 * i) * 256), 16, vo + (unsigned)i * 32u * ld * 4u, k0 * 4u, 0, 0);
            }
        } else {                                                        // kind B: rows 32 c .. 32 c + 31 of W1, all 256 k
            const unsigned c = cw;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const unsigned r = (unsigned)(wave + 4 * i);             // row in the slab
                const unsigned u = (unsigned)lane ^ (r & 15u);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW1, (__attribute__((address_space(3))) void*)(slot + r * 256), 16, (r * 256u + 4u * u) * 4u, c * 32u * 256u * 4u, 0, 0);
            }
        }
    };
    // fragment readers: 4 consecutive k of one weight row (this lane: row li of the 16-row tile, k group kg)
    // Addresses are one per-lane base plus a compile-time offset (the ds_read immediate), so that the 64 distinct fragment
    // addresses of a slab cost 6 registers, not 64: kind A's swizzle key ((16 t + li) >> 1) & 7 = (li >> 1) & 7 does not depend on
    // the tile; kind B's unit (4 t + kg) ^ li = (t >> 2) << 4 | ((t & 3) ^ (li >> 2)) << 2 | (kg ^ (li & 3)) needs one base per t & 3.
    int baseA[2], baseB[4];
#pragma unroll
    for (int h = 0; h < 2; ++h) baseA[h] = li * 32 + 4 * ((4 * h + kg) ^ ((li >> 1) & 7));
#pragma unroll
    for (int t3 = 0; t3 < 4; ++t3) baseB[t3] = li * 256 + 4 * (((t3 ^ (li >> 2)) << 2) | (kg ^ (li & 3)));
    auto fragA = [&](const float* slot, int t, int h) __attribute__((always_inline)) -> f32x4 {       // n tile t, k group h (16 k each)
        return *reinterpret_cast<const f32x4*>(slot + baseA[h] + t * 512);
    };
    auto fragB = [&](const float* slot, int j, int t) __attribute__((always_inline)) -> f32x4 {       // n2 tile j (0/1), k tile t
        return *reinterpret_cast<const f32x4*>(slot + baseB[t & 3] + j * 4096 + (t >> 2) * 64);
    };

    // ---- the attention output rows of this wave, in B-operand layout: lane (row li, kg) holds ao[row][k0 + 16 h + 4 kg .. + 3]
    const unsigned a_vo = ((unsigned)(m - blockIdx.x * XT_ROWS) * (unsigned)p.Kin + 4u * kg) * 4u;
    f32x4 fa[2], fn[2];                                                  // this slab's / the next slab's operand, [h]
    auto load_a = [&](int s) __attribute__((always_inline)) {
        fn[0] = bload(rsA, a_vo, (unsigned)s * 128u);
        fn[1] = bload(rsA, a_vo + 64u, (unsigned)s * 128u);
    };

    f32x4 x1[16], out[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) x1[t] = f32x4{0.f, 0.f, 0.f, 0.f};

    int s = 0;                                                           // slab counter (slot = s & 1)
    auto step = [&]() __attribute__((always_inline)) -> const float* {   // slab s has landed for every wave; start fetching s + 1
#ifndef XT_EXP_NOSYNC
        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");     // ... and slot (s + 1) & 1 is no longer read by anyone
        if (s + 1 < S) issue(s + 1);
#endif
        const float* slot = xt_sm + (s & 1) * XT_SLAB;
        ++s;
        return slot;
    };

    // ---- out-projection: x1^T[n][row] += Wo[n][k] ao[row][k], 32 k per slab
    issue(0);
    load_a(0);
    for (int i = 0; i < S1; ++i) {
        const float* slot = step();
        fa[0] = fn[0]; fa[1] = fn[1];
        if (i + 1 < S1) load_a(i + 1);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const f32x4 b = fa[h];
#pragma unroll
            for (int t = 0; t < 16; t += 2) {                            // two accumulators in turn (dependent latency 40 > issue 32 cycles)
                const f32x4 w0 = fragA(slot, t, h), w1 = fragA(slot, t + 1, h);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    x1[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(w0[r], b[r], x1[t], 0, 0, 0);
                    x1[t + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(w1[r], b[r], x1[t + 1], 0, 0, 0);
                }
            }
        }
        xt_pipeline<0>();
    }
    {                                                                    // + bias + residual: a lane holds 4 consecutive columns of each tile
        const float* rr = p.resid + (size_t)m * 256 + 4 * kg;
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            x1[t] += *reinterpret_cast<const f32x4*>(p.bo + 16 * t + 4 * kg) + *reinterpret_cast<const f32x4*>(rr + 16 * t);
            out[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }

    // ---- feed-forward, hidden columns in 16 chunks of 32 (two 16-column tiles)
    auto ff1 = [&](const float* slot, f32x4& h0, f32x4& h1) __attribute__((always_inline)) {       // K = 256 straight from the x1 accumulators
        h0 = f32x4{0.f, 0.f, 0.f, 0.f}; h1 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const f32x4 w0 = fragB(slot, 0, t), w1 = fragB(slot, 1, t);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                h0 = __builtin_amdgcn_mfma_f32_16x16x4f32(w0[r], x1[t][r], h0, 0, 0, 0);
                h1 = __builtin_amdgcn_mfma_f32_16x16x4f32(w1[r], x1[t][r], h1, 0, 0, 0);
            }
        }
    };
    auto ff2 = [&](const float* slot, const f32x4& hid0, const f32x4& hid1) __attribute__((always_inline)) {
        // out^T[n][row] += W2[n][32 c + k] hidden[row][k], k = 0 .. 31
#pragma unroll
        for (int t = 0; t < 16; t += 2) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const f32x4 w0 = fragA(slot, t, j), w1 = fragA(slot, t + 1, j);
                const f32x4 hv = j ? hid1 : hid0;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    out[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(w0[r], hv[r], out[t], 0, 0, 0);
                    out[t + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(w1[r], hv[r], out[t + 1], 0, 0, 0);
                }
            }
        }
        xt_pipeline<0>();
    };
    auto gelu_chunk = [&](int c, const f32x4& raw0, const f32x4& raw1, f32x4& hid0, f32x4& hid1) __attribute__((always_inline)) {
        const f32x4 b0 = *reinterpret_cast<const f32x4*>(p.b1 + 32 * c + 4 * kg), b1v = *reinterpret_cast<const f32x4*>(p.b1 + 32 * c + 16 + 4 * kg);
#pragma unroll
        for (int r = 0; r < 4; ++r) { hid0[r] = xt_gelu(raw0[r] + b0[r]); hid1[r] = xt_gelu(raw1[r] + b1v[r]); }
    };
    f32x4 raw0, raw1, nxt0, nxt1, hid0, hid1;
    {
        const float* slot = step();
        ff1(slot, raw0, raw1);
        xt_pipeline<0>();
    }
    for (int c = 0; c < 15; ++c) {
        {   // FF1 of chunk c + 1 on the matrix pipe with the GELU of chunk c (about 250 VALU instructions) between its MFMAs
            const float* slot = step();
            ff1(slot, nxt0, nxt1);
            gelu_chunk(c, raw0, raw1, hid0, hid1);
            xt_pipeline<2>();
        }
        ff2(step(), hid0, hid1);
        raw0 = nxt0; raw1 = nxt1;
    }
    gelu_chunk(15, raw0, raw1, hid0, hid1);
    ff2(step(), hid0, hid1);

    // ---- + b2 + x1, store: 16 bytes per lane and tile (a wave instruction covers 16 rows x 64 bytes)
    if (mvalid) {
        float* orow = p.out + (size_t)m * 256 + 4 * kg;
#pragma unroll
        for (int t = 0; t < 16; ++t)
            *reinterpret_cast<f32x4*>(orow + 16 * t) = out[t] + *reinterpret_cast<const f32x4*>(p.b2 + 16 * t + 4 * kg) + x1[t];
    }
}

hipError_t xf_tail_init() {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(&mocha_xf_tail), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * XT_SLAB * 4);
}

hipError_t launch_xf_tail(const XfTailParams& p, hipStream_t s) {
    if (p.M <= 0) return hipSuccess;
    if (p.Kin % 32 || p.Kin < 32 || (long long)XT_ROWS * p.Kin * 4 >= (1ll << 31) || 256ll * p.Kin * 4 >= (1ll << 31)) return hipErrorInvalidValue;
    hipLaunchKernelGGL(mocha_xf_tail, dim3((p.M + XT_ROWS - 1) / XT_ROWS), dim3(256), 2 * XT_SLAB * 4, s, p);
    return hipGetLastError();
}

}  // namespace mocha
