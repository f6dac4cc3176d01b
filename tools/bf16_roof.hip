// What the bf16 matrix pipe of this chip sustains on random operands (the roof mocha_gemm_x3 is priced against in DESIGN.md §5):
// a bare v_mfma_f32_32x32x16_bf16 loop with operands in registers, and the same loop with its operands re-read from LDS at the
// GEMM's rate (one ds_read_b128 per two MFMAs), three 256-thread workgroups per CU as in the GEMM, launched back to back for a
// few seconds so that the board reaches the clock it holds under that load.  MOCHA_ROOF_ZERO=1 fills the operands with zeros.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/bf16_roof.hip -o tools/bin/bf16_roof
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

template <bool LDS_OPERANDS>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void roof(const s16x8* __restrict__ src, float* __restrict__ out, int iters) {
    __shared__ __attribute__((aligned(16))) s16x8 sm[12 * 256];                  // 48 KB: three workgroups per CU
    const int tid = threadIdx.x;
    for (int i = tid; i < 12 * 256; i += 256) sm[i] = src[(blockIdx.x % 61) * 3072 + i];
    __syncthreads();
    s16x8 a[6], b[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) { a[i] = sm[i * 256 + tid]; b[i] = sm[(6 + i) * 256 + tid]; }
    f32x16 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
        if (LDS_OPERANDS) {                                                       // 12 reads per 24 MFMAs, as a K step of mocha_gemm_x3
#pragma unroll
            for (int i = 0; i < 6; ++i) { a[i] = sm[i * 256 + ((tid + it) & 255)]; b[i] = sm[(6 + i) * 256 + ((tid + 7 * it) & 255)]; }
        }
#pragma unroll
        for (int m = 0; m < 24; ++m)
            acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[m % 6], b[(m / 4) % 6], acc[m & 3], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + tid] = s;
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 2000;          // K steps per workgroup: 24 MFMAs per wave and step
    const double seconds = argc > 2 ? atof(argv[2]) : 2.0;
    const int wgs = 768 * 2;
    const bool zero = getenv("MOCHA_ROOF_ZERO") != nullptr;
    std::vector<unsigned short> h(61 * 3072 * 8);
    for (auto& v : h) {                                           // random bf16 in (-2, 2): random sign, exponent 126..127, random mantissa
        const unsigned r = (unsigned)rand();
        v = zero ? 0 : (unsigned short)(((r & 1) << 15) | ((126 + ((r >> 1) & 1)) << 7) | ((r >> 2) & 0x7f));
    }
    s16x8* d; float* o;
    CK(hipMalloc(&d, h.size() * 2)); CK(hipMalloc(&o, (size_t)wgs * 256 * 4));
    CK(hipMemcpy(d, h.data(), h.size() * 2, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int variant = 0; variant < 2; ++variant) {
        auto launch = [&]() {
            if (variant == 0) hipLaunchKernelGGL(roof<false>, dim3(wgs), dim3(256), 0, 0, d, o, iters);
            else hipLaunchKernelGGL(roof<true>, dim3(wgs), dim3(256), 0, 0, d, o, iters);
        };
        launch(); CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0, 0)); launch(); CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms1; CK(hipEventElapsedTime(&ms1, e0, e1));
        const int n = (int)(seconds * 1e3 / ms1) + 1;
        std::vector<float> window;
        CK(hipEventRecord(e0, 0));
        for (int i = 0; i < n; ++i) launch();
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        // the last tenth, after the board has settled
        const int tail = n / 10 + 1;
        CK(hipEventRecord(e0, 0));
        for (int i = 0; i < tail; ++i) launch();
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float mst; CK(hipEventElapsedTime(&mst, e0, e1));
        const double flops = (double)wgs * 4 * iters * 24 * (2.0 * 32 * 32 * 16);
        printf("%-46s first launch %7.1f TFLOP/s   %d launches over %.1f s: %7.1f TFLOP/s   settled: %7.1f TFLOP/s = %.2f of 2516.8\n",
               variant == 0 ? "bf16 MFMA, operands in registers" : "bf16 MFMA, 12 ds_read_b128 per 24 MFMAs",
               flops / (ms1 * 1e-3) / 1e12, n, ms * 1e-3, flops * n / (ms * 1e-3) / 1e12, flops * tail / (mst * 1e-3) / 1e12,
               flops * tail / (mst * 1e-3) / 1e12 / 2516.8);
    }
    return 0;
}
