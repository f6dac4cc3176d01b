// Device-side helpers shared by the .hip translation units (not included by host-only code).
#pragma once
#include <hip/hip_runtime.h>

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

}  // namespace mocha
