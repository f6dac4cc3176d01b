// Split-precision MFMA GEMM for gfx950: fp32-accurate results from the bf16 matrix pipe.
//
//   x = x1 + x2 + x3,  x1 = bf16(x), x2 = bf16(x - x1), x3 = bf16(x - x1 - x2)      (exact: 8+8+8 bits)
//   a*b = a1b1 + (a1b2 + a2b1) + (a1b3 + a3b1 + a2b2) + O(2^-24 |ab|)
//
// Six v_mfma_f32_32x32x16_bf16 (fp32 accumulate) per 32x32x16 block reproduce the fp32 product to
// ~2 ulp; the dropped terms a2b3, a3b2, a3b3 are below 2^-24 of the product.  The bf16 pipe is 16x
// the fp32 MFMA rate, so six passes are still 2.7x faster than one exact-f32 MFMA.  The same
// kernel with fewer planes serves the bf16-bank matcher: A (queries) in three planes, B (bank)
// in one -> exact fp32 queries against the bf16-rounded bank (3 MFMAs per block).
//
// Layout: activations stay fp32 in HBM; the A loader (plain rows or the temporal-conv gather of
// gemm_f32.hip) splits each value as it stages it into LDS.  Weights are split once on the host
// into [plane][N][K] bf16.  LDS rows are 32 bf16 + 8 pad (80 B): ds_read_b128 of the 8 consecutive
// k a lane feeds to the MFMA is conflict-free per 16-lane group (20*i mod 64 = 16 distinct
// multiples of 4).  Operands are swapped like in gemm_f32.hip (C^T accumulators, 16-B epilogue).
#include "kernels.h"
#include "device_utils.h"
#include <cstring>

namespace mocha {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

static constexpr int SBM = 128;
static constexpr int SBK = 32;                  // fp32 elements of K per slab = two 16-wide MFMA steps
static constexpr int SROW = 40;                 // bf16 per LDS row (32 + 8 pad) = 80 bytes

__device__ __forceinline__ float lrelu02s(float x) { return x > 0.f ? x : 0.2f * x; }
__device__ __forceinline__ float gelu_erfs(float x) { return 0.5f * x * (1.0f + mocha_erf(x * 0.70710678118654752440f)); }

// round-to-nearest-even bf16 of a finite float, as the high half-word
__device__ __forceinline__ unsigned bf16_rn(float x) {
    const unsigned u = __float_as_uint(x);
    return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}

// split 4 floats into NP bf16 planes; plane p of the 4 values packed into 8 bytes
template <int NP>
__device__ __forceinline__ void split4(const f32x4 v, u32x2 (&out)[NP]) {
    unsigned h[NP][4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        float r = v[e];
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            h[p][e] = bf16_rn(r);
            r = r - __uint_as_float(h[p][e] << 16);
        }
    }
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        out[p][0] = h[p][0] | (h[p][1] << 16);
        out[p][1] = h[p][2] | (h[p][3] << 16);
    }
}

// NPA / NPB: planes of the A (activation) and B (weight) operands.  Products kept: pa + pb <= MAXSUM (0-based)
template <int BN, int WM, int WN, int TM, int TN, int NPA, int NPB, int MAXSUM>
__global__ __launch_bounds__(256) void mocha_gemm_split(GemmParams p) {
    static_assert(WM * WN == 4 && WM * TM * 32 == SBM && WN * TN * 32 == BN, "tile shape");
    constexpr int NA = SBM / 32;                    // float4 loads of A per thread per slab (8 threads per row)
    constexpr int NBL = BN * 4 / 256;               // 16-byte loads per W plane per thread per slab (4 pieces per row)
    extern __shared__ __attribute__((aligned(16))) unsigned short smem_s[];
    unsigned short* As = smem_s;                            // [NPA][SBM][SROW]
    unsigned short* Bs = smem_s + NPA * SBM * SROW;         // [NPB][BN][SROW]

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hh = lane >> 5;
    const int wm = wave / WN, wn = wave % WN;

    const int n_tiles = (p.N + BN - 1) / BN;
    const int m_tiles = (p.M + SBM - 1) / SBM;
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
    const int m0 = mt * SBM, n0 = nt * BN;

    const int slabs_total = p.K / SBK;
    const int per = (slabs_total + p.ksplit - 1) / p.ksplit;
    const int s_begin = blockIdx.z * per;
    const int s_end = (s_begin + per) < slabs_total ? (s_begin + per) : slabs_total;

    // ---- A loader: 8 threads cover one 128-byte fp32 row segment (as in gemm_f32.hip)
    const int lrow = tid >> 3;
    const int lcol = (tid & 7) * 4;
    // operand fetches are buffer loads (device_utils.h): SGPR base, 32-bit lane offsets, scalar slab offset
    int a_rb[NA], a_t[NA];
    unsigned a_off[NA];
    const __amdgpu_buffer_rsrc_t rsA = make_rsrc(p.gather ? p.A : p.A + (size_t)(m0 < p.M ? m0 : 0) * p.lda);
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
    // ---- W loader: plane tile = BN rows x 64 bytes; 4 threads per row, 16 bytes each
    const int wr = tid >> 2;                        // 0..63
    const int wc = (tid & 3) * 8;                   // bf16 offset within the 32-wide slab
    unsigned w_off[NBL];
#pragma unroll
    for (int i = 0; i < NBL; ++i) {
        int n = n0 + wr + 64 * i;
        n = n < p.N ? n : p.N - 1;
        w_off[i] = ((unsigned)(n - n0) * (unsigned)p.K + wc) * 2u;
    }
    const size_t wplane = (size_t)p.N * p.K;
    __amdgpu_buffer_rsrc_t rsW[NPB];                 // one resource per plane, anchored at the tile's first row
#pragma unroll
    for (int pl = 0; pl < NPB; ++pl) rsW[pl] = make_rsrc(p.Wsplit + pl * wplane + (size_t)n0 * p.K);

    f32x4 ra[NA];
    u32x4 rb[NPB][NBL];
    auto load_slab = [&](int s) __attribute__((always_inline)) {
        const int k0 = s * SBK;
        if (!p.gather) {
#pragma unroll
            for (int i = 0; i < NA; ++i) ra[i] = bload(rsA, a_off[i], (unsigned)k0 * 4u);
        } else {
            const int tap = k0 / p.Cc;
            const unsigned c0 = (unsigned)(k0 - tap * p.Cc + lcol) * 4u;
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                int tf = a_t[i] * p.stride + tap - p.pad;
                tf = tf < 0 ? -tf : tf;
                tf = tf >= p.T_full ? 2 * (p.T_full - 1) - tf : tf;
                const int row = a_rb[i] + (tf >> p.tshift) * p.V;
                ra[i] = bload(rsA, (unsigned)row * (unsigned)p.lda * 4u + c0, 0u);
            }
        }
        if (p.a_lrelu) {
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                ra[i][0] = lrelu02s(ra[i][0]); ra[i][1] = lrelu02s(ra[i][1]);
                ra[i][2] = lrelu02s(ra[i][2]); ra[i][3] = lrelu02s(ra[i][3]);
            }
        }
#pragma unroll
        for (int pl = 0; pl < NPB; ++pl)
#pragma unroll
            for (int i = 0; i < NBL; ++i)
                rb[pl][i] = __builtin_bit_cast(u32x4, bload(rsW[pl], w_off[i], (unsigned)k0 * 2u));
    };
    auto store_slab = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            u32x2 pl[NPA];
            split4<NPA>(ra[i], pl);
#pragma unroll
            for (int q = 0; q < NPA; ++q)
                *reinterpret_cast<u32x2*>(As + (q * SBM + lrow + 32 * i) * SROW + lcol) = pl[q];
        }
#pragma unroll
        for (int pl = 0; pl < NPB; ++pl)
#pragma unroll
            for (int i = 0; i < NBL; ++i)
                *reinterpret_cast<u32x4*>(Bs + (pl * BN + wr + 64 * i) * SROW + wc) = rb[pl][i];
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

    // operand fragment of lane (row l31, k-half hh) for MFMA step ks: 8 consecutive bf16 at k = 16*ks + 8*hh
    const unsigned short* Ab = As + (wm * TM * 32 + l31) * SROW + 8 * hh;
    const unsigned short* Bb = Bs + (wn * TN * 32 + l31) * SROW + 8 * hh;
    for (int s = s_begin; s < s_end; ++s) {
        const bool more = (s + 1) < s_end;
        if (more) load_slab(s + 1);
#pragma unroll
        for (int ks = 0; ks < SBK / 16; ++ks) {
            s16x8 a[NPA][TM], b[NPB][TN];
#pragma unroll
            for (int q = 0; q < NPA; ++q)
#pragma unroll
                for (int i = 0; i < TM; ++i)
                    a[q][i] = *reinterpret_cast<const s16x8*>(Ab + (q * SBM + i * 32) * SROW + 16 * ks);
#pragma unroll
            for (int q = 0; q < NPB; ++q)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    b[q][j] = *reinterpret_cast<const s16x8*>(Bb + (q * BN + j * 32) * SROW + 16 * ks);
            // low-order products first, the dominant a1*b1 last
#pragma unroll
            for (int sum = MAXSUM; sum >= 0; --sum)
#pragma unroll
                for (int pa = 0; pa < NPA; ++pa) {
                    const int pb = sum - pa;
                    if (pb < 0 || pb >= NPB) continue;
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[pb][j], a[pa][i], acc[i][j], 0, 0, 0);   // C^T tile
                }
        }
        __syncthreads();
        if (more) store_slab();
        __syncthreads();
    }

    // ---- epilogue (identical to gemm_f32.hip): lane&31 = row, regs 4g..4g+3 = 4 consecutive columns
    float* Cz = p.C + (size_t)blockIdx.z * p.slab_stride;
    const bool raw = p.ksplit > 1;
    const bool vec_ok = ((p.ldc & 3) == 0) && (!p.residual || (p.ldr & 3) == 0) && ((p.N & 3) == 0 || raw);
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int row = m0 + (wm * TM + i) * 32 + l31;
        if (row >= p.M) continue;
        const float* rbrow = (!raw && p.rowbias) ? p.rowbias + (size_t)(row % p.rb_mod) * p.N : nullptr;
        const float* rsrow = (!raw && p.residual) ? p.residual + (size_t)row * p.ldr : nullptr;
        float* crow = Cz + (size_t)row * p.ldc;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int col = n0 + (wn * TN + j) * 32 + 8 * g + 4 * hh;
                if (col >= p.N) continue;
                f32x4 v = {acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
                if (vec_ok && col + 3 < p.N) {
                    if (!raw) {
                        if (p.bias) v += *reinterpret_cast<const f32x4*>(p.bias + col);
                        if (rbrow) v += *reinterpret_cast<const f32x4*>(rbrow + col);
                        if (p.act == 1) { v[0] = gelu_erfs(v[0]); v[1] = gelu_erfs(v[1]); v[2] = gelu_erfs(v[2]); v[3] = gelu_erfs(v[3]); }
                        else if (p.act == 2) { v[0] = lrelu02s(v[0]); v[1] = lrelu02s(v[1]); v[2] = lrelu02s(v[2]); v[3] = lrelu02s(v[3]); }
                        else if (p.act == 3) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f); }
                        if (rsrow) v += *reinterpret_cast<const f32x4*>(rsrow + col);
                    }
                    *reinterpret_cast<f32x4*>(crow + col) = v;
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int c1 = col + e;
                        if (c1 >= p.N) continue;
                        float x = v[e];
                        if (!raw) {
                            if (p.bias) x += p.bias[c1];
                            if (rbrow) x += rbrow[c1];
                            if (p.act == 1) x = gelu_erfs(x);
                            else if (p.act == 2) x = lrelu02s(x);
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

template <int BN, int NPA, int NPB>
static constexpr size_t split_lds() { return (size_t)(NPA * SBM + NPB * BN) * SROW * sizeof(unsigned short); }

hipError_t gemm_split_init() {
    hipError_t e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(&mocha_gemm_split<128, 2, 2, 2, 2, 3, 3, 2>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)split_lds<128, 3, 3>());
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(&mocha_gemm_split<64, 4, 1, 1, 2, 3, 3, 2>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)split_lds<64, 3, 3>());
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(&mocha_gemm_split<128, 2, 2, 2, 2, 3, 1, 2>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)split_lds<128, 3, 1>());
    return e;
}

// planes: 33 = activations and weights in 3 planes (fp32-accurate GEMM); 31 = 3 query planes x 1 bank plane
hipError_t launch_gemm_split(const GemmParams& p, int planes, hipStream_t s) {
    if (p.M <= 0 || p.N <= 0) return hipSuccess;
    if (p.K % SBK != 0 || !p.Wsplit) return hipErrorInvalidValue;
    if (p.gather && ((p.Cc % SBK != 0) || p.R != 1)) return hipErrorInvalidValue;
    const int m_tiles = (p.M + SBM - 1) / SBM;
    const int m_pad = m_tiles >= 8 ? (m_tiles + 7) / 8 * 8 : m_tiles;
    if (planes == 31) {
        dim3 grid(m_pad * ((p.N + 127) / 128), 1, p.ksplit);
        constexpr size_t lds = split_lds<128, 3, 1>();
        hipLaunchKernelGGL((mocha_gemm_split<128, 2, 2, 2, 2, 3, 1, 2>), grid, dim3(256), lds, s, p);
    } else if (gemm_is_narrow(p)) {
        dim3 grid(m_pad * ((p.N + 63) / 64), 1, p.ksplit);
        constexpr size_t lds = split_lds<64, 3, 3>();
        hipLaunchKernelGGL((mocha_gemm_split<64, 4, 1, 1, 2, 3, 3, 2>), grid, dim3(256), lds, s, p);
    } else {
        dim3 grid(m_pad * ((p.N + 127) / 128), 1, p.ksplit);
        constexpr size_t lds = split_lds<128, 3, 3>();
        hipLaunchKernelGGL((mocha_gemm_split<128, 2, 2, 2, 2, 3, 3, 2>), grid, dim3(256), lds, s, p);
    }
    return hipGetLastError();
}

// host-side 3-plane split of a weight matrix: out[plane][n][k] bf16 (same rounding as the device loader)
void split_weights_host(const float* w, size_t count, unsigned short* out /*3*count*/) {
    auto rn = [](float x) -> unsigned short {
        unsigned u; memcpy(&u, &x, 4);
        return (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
    };
    auto up = [](unsigned short h) -> float { unsigned u = (unsigned)h << 16; float f; memcpy(&f, &u, 4); return f; };
    for (size_t i = 0; i < count; ++i) {
        float r = w[i];
        for (int pl = 0; pl < 3; ++pl) {
            const unsigned short h = rn(r);
            out[pl * count + i] = h;
            r = r - up(h);
        }
    }
}

}  // namespace mocha
