// TEST-ONLY canary for tests/test_concurrency_stress.py: the ROUND-2 build of mocha_body_front (pointwise.hip), kept outside the
// product library.  Its 72 adjacency coefficients are read from LDS where they are used and the ext-vector arithmetic compiles to
// v_pk_fma_f32 with op_sel on the freshly returned registers; beside another stream's bf16-MFMA-plus-VALU kernels (mocha_gemm_x3)
// it intermittently lost terms (lanes 48-63 of a result exactly 0; profiles/r02/f_body_front_repro.txt, tools/body_front_repro.hip).
// The stress test runs it through the same harness as the shipped kernels to show that the harness still detects that failure.
//   out[(f,w)][k*256+c] = sum_v A_b[k][v][w] lrelu(x[(f,v)][c])          (net/blocks.py:131, :64)
#include <hip/hip_runtime.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float lrelu02(float x) { return x > 0.f ? x : 0.2f * x; }

__global__ __launch_bounds__(256) void canary_body_front(const float* __restrict__ x, const float* __restrict__ Ab, float* __restrict__ out, int frames) {
    __shared__ float a[72];
    if (threadIdx.x < 72) a[threadIdx.x] = Ab[threadIdx.x];
    __syncthreads();
    const int gid = blockIdx.x * 256 + threadIdx.x;
    const int f = gid >> 6, c4 = (gid & 63) * 4;
    if (f >= frames) return;
    f32x4 xv[6];
#pragma unroll
    for (int v = 0; v < 6; ++v) {
        f32x4 t = *reinterpret_cast<const f32x4*>(x + ((size_t)f * 6 + v) * 256 + c4);
        t[0] = lrelu02(t[0]); t[1] = lrelu02(t[1]); t[2] = lrelu02(t[2]); t[3] = lrelu02(t[3]);
        xv[v] = t;
    }
#pragma unroll
    for (int w = 0; w < 6; ++w)
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int v = 0; v < 6; ++v) acc += xv[v] * a[(k * 6 + v) * 6 + w];
            *reinterpret_cast<f32x4*>(out + ((size_t)f * 6 + w) * 512 + k * 256 + c4) = acc;
        }
}

extern "C" int canary_body_front_launch(const float* x, const float* Ab, float* out, int frames, void* stream) {
    if (frames <= 0) return 0;
    const long long threads = (long long)frames * 64;
    hipLaunchKernelGGL(canary_body_front, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, Ab, out, frames);
    return (int)hipGetLastError();
}
