// Multi-head softmax attention over <= 96 tokens on gfx950, exact-f32 MFMA
// (reference: Attention.forward, net/transformer.py:65-76).
//
// One workgroup of NQW waves per (window, head); wave w owns query block 32w..32w+31; keys are padded to
// NKT tiles of 32.  <DH,3,3> serves the Generator (90 tokens, head dim 128 / 256); <64,6,6>, <64,6,3>,
// <64,3,3> serve the CVAE sampler (182-token prior, 90 queries x 181 memory, 90 x 90; model_CVAE.py).
//   1. S^T = K · Q^T  (keys on the MFMA rows, queries on the lanes): with the query on the
//      lane, the softmax over keys is a per-lane reduction over the accumulator registers
//      plus one cross-half shuffle — no LDS round trip, no 32-lane butterflies.
//   2. softmax in registers (max-subtracted, exact expf, division by the row sum).
//   3. O^T = V^T · P^T: the accumulator tile P^T (keys in registers, queries on lanes) IS
//      the B operand of v_mfma_f32_32x32x2_f32 register by register (k = key), so P never
//      leaves the register file.  V is staged through LDS 64 head-dims at a time.
// Tokens are padded 90 -> 96 with zero rows; padded keys are masked to -inf.
#include "kernels.h"
#include "device_utils.h"

namespace mocha {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

static constexpr int LK = 36;       // LDS row stride for the K/Q chunks (32 + 4 pad)
static constexpr int DV = 64;       // head dims of V staged per pass

// Occupancy is pinned per instance: left alone, hipcc gives the dh = 256 instance 188 registers (2 waves per SIMD); at 4 waves
// (128 VGPRs, accumulators in VGPRs, 28 bytes of spill outside the MFMA loops) it runs 0.53 -> 0.44 ms per demo step.
template <int DH, int NKT, int NQW>
__global__ __launch_bounds__(NQW * 64) __attribute__((amdgpu_waves_per_eu(NKT != 3 ? 1 : (DH == 128 ? 3 : 4)))) void mocha_attention_f32(AttnParams p) {
    constexpr int NTHR = NQW * 64;
    constexpr int KROWS = NKT * 32, QROWS = NQW * 32;
    constexpr int KPT = KROWS * 8 / NTHR;           // float4 of a K chunk per thread
    constexpr int QPT = QROWS * 8 / NTHR;           // = 4
    constexpr int VPT = KROWS * 16 / NTHR;          // float4 of a V pass per thread
    static_assert(KROWS * 8 % NTHR == 0 && KROWS * 16 % NTHR == 0 && DH % DV == 0, "shape");
    constexpr int SMF = (KROWS + QROWS) * LK > KROWS * DV ? (KROWS + QROWS) * LK : KROWS * DV;
    __shared__ __attribute__((aligned(16))) float smem[SMF];           // V pass reuses the K/Q stage
    float* Ks = smem;
    float* Qs = smem + KROWS * LK;
    float* Vs = smem;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hh = lane >> 5;
    // XCD-aware order (workgroup id % 8 = XCD): the heads of one window run on the same XCD, so keys / values that the heads
    // share (the folded decoder: IN(cha) and cha for all four heads) are fetched from HBM once and then hit that XCD's L2
    const int id = blockIdx.x;
    const int slot = id >> 3;
    const int head = slot % p.heads;
    const int b = (slot / p.heads) * 8 + (id & 7);
    if (b >= p.B) return;
    const int nq = p.nq, nk = p.nk;

    const float* qg = p.q + (size_t)b * nq * p.ldq + head * DH;
    const float* kg = p.k + (size_t)b * nk * p.ldk + head * (p.hsk < 0 ? DH : p.hsk);
    const float* vg = p.v + (size_t)b * nk * p.ldv + head * (p.hsv < 0 ? DH : p.hsv);
    // operand fetches as buffer loads: (window, head) base in SGPRs, 32-bit lane offsets (<= 192 rows x ld x 4 B)
    const __amdgpu_buffer_rsrc_t rsq = make_rsrc(qg), rsk = make_rsrc(kg), rsv = make_rsrc(vg);

    // ---------------- phase 1: S^T[key][query] over DH in chunks of 32
    f32x16 st[NKT];
#pragma unroll
    for (int t = 0; t < NKT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) st[t][r] = 0.f;

    // chunk c+1 of K and Q is fetched into registers while chunk c is multiplied (the global-load
    // latency of a chunk is about the length of its 48 MFMAs, so an un-prefetched loop idles half the time)
    f32x4 kr[KPT], qr[QPT];
    auto fetch_kq = [&](int c) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < KPT; ++i) {         // rows x 8 float4 per chunk
            const int f = tid + NTHR * i;
            const int row = f >> 3, c4 = (f & 7) * 4;
            f32x4 z = {0.f, 0.f, 0.f, 0.f};
            kr[i] = row < nk ? bload(rsk, (unsigned)(row * p.ldk + c4) * 4u, (unsigned)c * 128u) : z;
        }
#pragma unroll
        for (int i = 0; i < QPT; ++i) {
            const int f = tid + NTHR * i;
            const int row = f >> 3, c4 = (f & 7) * 4;
            f32x4 z = {0.f, 0.f, 0.f, 0.f};
            qr[i] = row < nq ? bload(rsq, (unsigned)(row * p.ldq + c4) * 4u, (unsigned)c * 128u) : z;
        }
    };
    auto stage_kq = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < KPT; ++i) {
            const int f = tid + NTHR * i;
            *reinterpret_cast<f32x4*>(Ks + (f >> 3) * LK + (f & 7) * 4) = kr[i];
        }
#pragma unroll
        for (int i = 0; i < QPT; ++i) {
            const int f = tid + NTHR * i;
            *reinterpret_cast<f32x4*>(Qs + (f >> 3) * LK + (f & 7) * 4) = qr[i];
        }
    };
    fetch_kq(0);
    stage_kq();
    __syncthreads();
    for (int c = 0; c < DH / 32; ++c) {
        if (c + 1 < DH / 32) fetch_kq(c + 1);
#pragma unroll
        for (int kgp = 0; kgp < 4; ++kgp) {
            f32x4 a[NKT];
#pragma unroll
            for (int t = 0; t < NKT; ++t) a[t] = *reinterpret_cast<const f32x4*>(Ks + (t * 32 + l31) * LK + kgp * 8 + 4 * hh);
            const f32x4 bq = *reinterpret_cast<const f32x4*>(Qs + (wave * 32 + l31) * LK + kgp * 8 + 4 * hh);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                for (int t = 0; t < NKT; ++t)
                    st[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t][ks], bq[ks], st[t], 0, 0, 0);
        }
        __syncthreads();
        if (c + 1 < DH / 32) stage_kq();
        __syncthreads();
    }

    // ---------------- phase 2: softmax over keys for this lane's query
    // st[t][r] = S[query = 32*wave + l31][key = 32t + (r&3) + 8(r>>2) + 4hh]
    // masking is only needed in tiles that can hold padded keys; scores stay unscaled until the exponent:
    // softmax(scale * s) = exp2((s - max s) * scale * log2 e) / sum   (scale > 0), one fma + v_exp_f32 per element
    float mx = -INFINITY;
#pragma unroll
    for (int t = 0; t < NKT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = t * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
            if (key >= nk) st[t][r] = -INFINITY;              // uniform per (t, r, half): cheap select, no effect on valid keys
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
            const float e = __builtin_amdgcn_exp2f(fmaf(st[t][r], c2, mb));     // exp2(-inf) = 0 for masked keys
            st[t][r] = e;
            sum += e;
        }
    sum += __shfl_xor(sum, 32);
    const float inv = 1.0f / sum;                   // one IEEE division per query, then 16*NKT multiplies (<= 1 ulp from e / sum)
#pragma unroll
    for (int t = 0; t < NKT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) st[t][r] = st[t][r] * inv;

    // ---------------- phase 3: O^T[d][query] = sum_key V[key][d] * P^T[key][query]
    const int query = wave * 32 + l31;
    float* og = p.out + ((size_t)b * nq + query) * p.ldo + head * DH;
    // V is staged 64 head-dims at a time; pass dp+1 is fetched into registers during pass dp
    f32x4 vr[VPT];
    auto fetch_v = [&](int dp) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < VPT; ++i) {         // rows x 16 float4 per pass
            const int f = tid + NTHR * i;
            const int row = f >> 4, c4 = (f & 15) * 4;
            f32x4 z = {0.f, 0.f, 0.f, 0.f};
            vr[i] = row < nk ? bload(rsv, (unsigned)(row * p.ldv + c4) * 4u, (unsigned)dp * (DV * 4u)) : z;
        }
    };
    auto stage_v = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < VPT; ++i) {
            const int f = tid + NTHR * i;
            const int row = f >> 4, c4 = (f & 15) * 4;
            *reinterpret_cast<f32x4*>(Vs + row * DV + c4) = vr[i];
        }
    };
    fetch_v(0);                                  // (issued before the softmax above would be even better; the
    stage_v();                                   //  K/Q stage is free: every wave passed the last barrier of phase 1)
    __syncthreads();
    for (int dp = 0; dp < DH / DV; ++dp) {
        if (dp + 1 < DH / DV) fetch_v(dp + 1);
        f32x16 o[2];
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[d][r] = 0.f;
#pragma unroll
        for (int t = 0; t < NKT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = t * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
#pragma unroll
                for (int d = 0; d < 2; ++d) {
                    const float av = Vs[key * DV + d * 32 + l31];      // A[i = d-col][k = key]
                    o[d] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, st[t][r], o[d], 0, 0, 0);
                }
            }
        // o[d][r] = O[query][dcol = dp*64 + d*32 + (r&3) + 8(r>>2) + 4hh]: regs 4g..4g+3 are 4 consecutive dims
        if (query < nq) {
#pragma unroll
            for (int d = 0; d < 2; ++d)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    f32x4 w = {o[d][4 * g], o[d][4 * g + 1], o[d][4 * g + 2], o[d][4 * g + 3]};
                    *reinterpret_cast<f32x4*>(og + dp * DV + d * 32 + 8 * g + 4 * hh) = w;
                }
        }
        __syncthreads();
        if (dp + 1 < DH / DV) stage_v();
        __syncthreads();
    }
}

hipError_t launch_attention(const AttnParams& p, hipStream_t s) {
    if (p.B <= 0) return hipSuccess;
    if (p.nq < 1 || p.nk < 1 || p.nq > 192 || p.nk > 192) return hipErrorInvalidValue;
    dim3 grid((unsigned)(((p.B + 7) / 8) * 8 * p.heads));
    const bool small = p.nq <= 96 && p.nk <= 96;
    if (p.dh == 128 && small) hipLaunchKernelGGL((mocha_attention_f32<128, 3, 3>), grid, dim3(192), 0, s, p);
    else if (p.dh == 256 && small) hipLaunchKernelGGL((mocha_attention_f32<256, 3, 3>), grid, dim3(192), 0, s, p);
    else if (p.dh == 64 && small) hipLaunchKernelGGL((mocha_attention_f32<64, 3, 3>), grid, dim3(192), 0, s, p);
    else if (p.dh == 64 && p.nq <= 96) hipLaunchKernelGGL((mocha_attention_f32<64, 6, 3>), grid, dim3(192), 0, s, p);
    else if (p.dh == 64) hipLaunchKernelGGL((mocha_attention_f32<64, 6, 6>), grid, dim3(384), 0, s, p);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

}  // namespace mocha
