// Device-side helpers shared by the .hip translation units (not included by host-only code).
#pragma once
#include <hip/hip_runtime.h>

// Kernels that must not contain packed fp32 instructions (v_pk_fma_f32 ...): see mocha_body_front in pointwise.hip.  The attribute
// only means something to the device pass.
#if defined(__HIP_DEVICE_COMPILE__)
#define MOCHA_NO_PACKED_F32 __attribute__((target("no-packed-fp32-ops")))
#else
#define MOCHA_NO_PACKED_F32
#endif

namespace mocha {

typedef float f32x4_t __attribute__((ext_vector_type(4)));

// 16-byte buffer load: base address in SGPRs (a buffer resource), per-lane 32-bit byte offset, scalar byte offset.
// Measured (tools/mfma_probe.hip): in an MFMA K loop the same fetches as global_load_dwordx4 with 64-bit VGPR addresses
// cost 11-17 % of the matrix pipe (a 64-bit VALU add per load on top of the load's issue), as buffer loads 6-13 %.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, 0x7fffffff, 0x00020000);   // raw, 2 GiB window, no swizzle
}
__device__ __forceinline__ f32x4_t bload(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(f32x4_t, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}

__device__ __forceinline__ void bstore(__amdgpu_buffer_rsrc_t r, f32x4_t v, unsigned voff, unsigned soff) {
    typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, v), r, voff, soff, 0);
}
template <int AUX>      // cache policy bits as a compile-time constant (2 = nt)
__device__ __forceinline__ void bstore_aux(__amdgpu_buffer_rsrc_t r, f32x4_t v, unsigned voff, unsigned soff) {
    typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, v), r, voff, soff, AUX);
}

// erf to < 1 ulp (5.8e-8 absolute), branch-free: both ranges are evaluated and selected - straight-line VALU code about half
// as long as the library erff with its per-lane branches (the GELU of net/transformer.py:27 sits in GEMM epilogues).  Coefficients: N. Juffa's single-precision erff (two minimax polynomials, split at 0.9277); checked against a float64
// erf over [-6, 6] (tests/test_erf_polynomial.py).
__device__ __forceinline__ float mocha_erf(float a) {
    const float t = fabsf(a), s = a * a;
    float r = fmaf(-1.72853470e-5f, t, 3.83197126e-4f);
    const float u = fmaf(-3.88396438e-3f, t, 2.42546219e-2f);
    r = fmaf(r, s, u);
    r = fmaf(r, t, -1.06777877e-1f);
    r = fmaf(r, t, -6.34846687e-1f);
    r = fmaf(r, t, -1.28717512e-1f);
    r = fmaf(r, t, -t);
    const float big = copysignf(1.0f - __builtin_amdgcn_exp2f(r * 1.44269504088896340736f), a);
    float q = -5.96761703e-4f;
    q = fmaf(q, s, 4.99119423e-3f);
    q = fmaf(q, s, -2.67681349e-2f);
    q = fmaf(q, s, 1.12819925e-1f);
    q = fmaf(q, s, -3.76125336e-1f);
    q = fmaf(q, s, 1.28379166e-1f);
    const float small = fmaf(q, a, a);
    return t > 0.927734375f ? big : small;
}

// The exact-erf GELU of net/transformer.py:27 (nn.GELU()) for the GEMM epilogues: GELU(x) = x Phi(x), Phi(x) = 0.5 erfc(-x / sqrt 2).
// With t = |x| / sqrt 2:  h = 0.5 erfc(t) = exp2(q(t)), q a degree-9 polynomial fitted to log2(0.5 erfc) on [0, 4.3] with weight erfc
// (the error of h is what counts: 2e-9 from the fit, the rest is float32 evaluation), h = 0 beyond (0.5 erfc < 6e-10);
// Phi = h for x < 0 (no cancellation on the negative tail), 1 - h otherwise.  One polynomial and one v_exp_f32 per element - 17 + 4 issue
// slots against the 26 + 4 of 0.5 x (1 + mocha_erf(x / sqrt 2)) with its two polynomials - and |error| <= 1.1e-7 max(1, |x|), relative error
// below 2.2e-6 for x > -3 (the erf form: 1.1e-7 and 1.8e-5).  Checked against float64 in tests/test_erf_polynomial.py.
__device__ __forceinline__ float mocha_gelu(float x) {
    const float t = fabsf(x) * 0.70710678118654752440f;
    float q = 1.146809295e-05f;
    q = fmaf(q, t, -1.515590512e-04f);
    q = fmaf(q, t, 8.423155240e-04f);
    q = fmaf(q, t, -2.261537520e-03f);
    q = fmaf(q, t, 6.770915453e-05f);
    q = fmaf(q, t, 2.773738608e-02f);
    q = fmaf(q, t, -1.483134404e-01f);
    q = fmaf(q, t, -9.184416673e-01f);
    q = fmaf(q, t, -1.627907386e+00f);
    q = fmaf(q, t, -9.999999969e-01f);
    float h = __builtin_amdgcn_exp2f(q);
    h = t > 4.3f ? 0.f : h;
    return x * (x < 0.f ? h : 1.0f - h);
}

// four at a time, the polynomial on float pairs (v_pk_fma_f32 / v_pk_mul_f32: half the instructions); the same operations per element
// as mocha_gelu, so the results are the same bits
__device__ __forceinline__ f32x4_t mocha_gelu4(f32x4_t x) {
    typedef float f2 __attribute__((ext_vector_type(2)));
    f32x4_t out;
#pragma unroll
    for (int hpair = 0; hpair < 2; ++hpair) {
        const f2 xv = {x[2 * hpair], x[2 * hpair + 1]};
        const f2 t = __builtin_elementwise_abs(xv) * 0.70710678118654752440f;
        f2 q = {1.146809295e-05f, 1.146809295e-05f};
        q = __builtin_elementwise_fma(q, t, (f2){-1.515590512e-04f, -1.515590512e-04f});
        q = __builtin_elementwise_fma(q, t, (f2){8.423155240e-04f, 8.423155240e-04f});
        q = __builtin_elementwise_fma(q, t, (f2){-2.261537520e-03f, -2.261537520e-03f});
        q = __builtin_elementwise_fma(q, t, (f2){6.770915453e-05f, 6.770915453e-05f});
        q = __builtin_elementwise_fma(q, t, (f2){2.773738608e-02f, 2.773738608e-02f});
        q = __builtin_elementwise_fma(q, t, (f2){-1.483134404e-01f, -1.483134404e-01f});
        q = __builtin_elementwise_fma(q, t, (f2){-9.184416673e-01f, -9.184416673e-01f});
        q = __builtin_elementwise_fma(q, t, (f2){-1.627907386e+00f, -1.627907386e+00f});
        q = __builtin_elementwise_fma(q, t, (f2){-9.999999969e-01f, -9.999999969e-01f});
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            float h = __builtin_amdgcn_exp2f(q[e]);
            h = t[e] > 4.3f ? 0.f : h;
            out[2 * hpair + e] = xv[e] * (xv[e] < 0.f ? h : 1.0f - h);
        }
    }
    return out;
}

// ---------------------------------------------------------------------------------------------------------------------
// Plane split (gemm_x3.hip, attention_x3.hip, pointwise.hip): x = x0 + x1 + x2 with x0 = bf16(x), x1 = bf16(x - x0),
// x2 = bf16(x - x0 - x1), round to nearest even (v_cvt_pk_bf16_f32).  Exact for practically every fp32 value, residual <= 2^-24 |x|
// otherwise; bf16 x bf16 products are exact in fp32, so six MFMA passes reproduce an fp32 product to within one fp32 rounding
// (tests/test_plane_split_numerics.py).
// ---------------------------------------------------------------------------------------------------------------------
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
typedef short s16x8_t __attribute__((ext_vector_type(8)));

__device__ __forceinline__ unsigned cvt_pk_bf16(float a, float b) {
    const f32x2_t v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));       // low half = bf16(a)
}
// four floats -> three planes of four bf16 (8 bytes each)
__device__ __forceinline__ void plane_split4(const f32x4_t v, u32x2_t (&out)[3]) {
    float r0 = v[0], r1 = v[1], r2 = v[2], r3 = v[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        const unsigned p01 = cvt_pk_bf16(r0, r1), p23 = cvt_pk_bf16(r2, r3);
        out[q][0] = p01; out[q][1] = p23;
        if (q < 2) {
            r0 -= __uint_as_float(p01 << 16); r1 -= __uint_as_float(p01 & 0xffff0000u);
            r2 -= __uint_as_float(p23 << 16); r3 -= __uint_as_float(p23 & 0xffff0000u);
        }
    }
}
// eight floats -> three planes of eight bf16: one MFMA operand (K = 16, this lane's k half) per plane
__device__ __forceinline__ void plane_split8(const float (&x)[8], s16x8_t (&out)[3]) {
    const f32x4_t lo = {x[0], x[1], x[2], x[3]}, hi = {x[4], x[5], x[6], x[7]};
    u32x2_t a[3], b[3];
    plane_split4(lo, a); plane_split4(hi, b);
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        const u32x4_t v = {a[q][0], a[q][1], b[q][0], b[q][1]};
        out[q] = __builtin_bit_cast(s16x8_t, v);
    }
}
// fp32 -> bf16 bits, round to nearest even; a NaN stays a NaN (v_cvt_pk_bf16_f32).  ONE definition: the matcher's bounds rely on the
// centred bf16 query plane being rounded identically wherever it is produced (mocha_center_bf16, mocha_instnorm's zc16) and measured
// (mocha_match_select's ||dq||)
__device__ __forceinline__ unsigned bf16_bits(float x) { return (unsigned)__builtin_bit_cast(unsigned short, (__bf16)x); }

// the order the six products a_i b_j (i + j <= 2) are accumulated in: low-order ones first, a0 b0 last
static constexpr int PLANE_PA[6] = {0, 1, 2, 0, 1, 0}, PLANE_PB[6] = {2, 1, 0, 1, 0, 0};

}  // namespace mocha
