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

}  // namespace mocha
