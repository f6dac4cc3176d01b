// In-order execution within a stream while other streams keep the device busy: NS streams, each running a chain
//     produce(u, tag) -> consume(u, tag)          (consume counts elements that do not carry the tag its own stream just wrote)
// plus, on the odd streams, an aggressor between the two (a mid-size mocha_gemm_x3, or with MOCHA_ORDER_SPIN=1 a plain spinning
// kernel with the same launch shape: 256 threads, 52 KB of dynamic LDS).  With more streams than hardware queues (HIP multiplexes
// its streams over GPU_MAX_HW_QUEUES = 4 of them) some streams share a queue.  Any non-zero count is an ordering violation
// inside ONE stream.  Written to narrow down the intermittent wrong rows of the two-context test (tools/experiments/README.md).
// build: tools/build_gemm_bench.sh, then
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I mocha_sigasia2023_amd/csrc -c tools/stream_order.hip -o tools/bin/stream_order.o && hipcc --offload-arch=gfx950 tools/bin/stream_order.o mocha_sigasia2023_amd/csrc/gemm_f32.o mocha_sigasia2023_amd/csrc/gemm_x3.o -o tools/bin/stream_order
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "kernels.h"
using namespace mocha;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

// every thread writes `per` floats, slowly (a dependent chain between stores), so that the kernel's tail is long
__global__ void produce(float* u, int n, int per, float tag, int delay) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    float x = tag;
    for (int j = 0; j < per; ++j) {
        for (int d = 0; d < delay; ++d) x = __builtin_fmaf(x, 1.0f, 0.0f) + 0.0f * (float)d;
        const int i = j * (gridDim.x * blockDim.x) + t;
        if (i < n) u[i] = x;
    }
}
__global__ void consume(const float* u, int n, float tag, unsigned long long* bad) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned c = 0;
    for (int i = t; i < n; i += gridDim.x * blockDim.x) c += u[i] != tag;
    if (c) atomicAdd(bad, (unsigned long long)c);
}
__global__ void spin(float* out, int iters) {
    extern __shared__ float sm[];
    float x = (float)threadIdx.x;
    for (int i = 0; i < iters; ++i) { sm[(threadIdx.x + i) & 8191] = x; x = sm[(threadIdx.x * 7 + i) & 8191] + 1.0f; }
    if (x == -1.f) out[0] = x;
}

int main(int argc, char** argv) {
    const int NS = argc > 1 ? atoi(argv[1]) : 8, reps = argc > 2 ? atoi(argv[2]) : 300;
    const bool use_spin = getenv("MOCHA_ORDER_SPIN") != nullptr, prio = getenv("MOCHA_ORDER_PRIO") != nullptr;
    CK(gemm_init()); CK(gemm_x3_init());
    const int n = 27000 * 1280 / 8;                                // a few MB per stream
    std::vector<hipStream_t> st(NS); std::vector<float*> u(NS); std::vector<unsigned long long*> bad(NS);
    int lo, hi; CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
    for (int i = 0; i < NS; ++i) {
        if (prio) CK(hipStreamCreateWithPriority(&st[i], hipStreamNonBlocking, (i & 1) ? hi : lo));
        else CK(hipStreamCreateWithFlags(&st[i], hipStreamNonBlocking));
        CK(hipMalloc(&u[i], (size_t)n * 4)); CK(hipMalloc(&bad[i], 8)); CK(hipMemset(bad[i], 0, 8));
    }
    // aggressor operands
    const int M = 6750, N = 512, K = 256;
    float *A, *W, *C; unsigned short* Wp;
    CK(hipMalloc(&A, (size_t)M * K * 4)); CK(hipMalloc(&W, (size_t)N * K * 4)); CK(hipMalloc(&C, (size_t)NS * M * N * 4));
    CK(hipMemset(A, 0, (size_t)M * K * 4)); CK(hipMemset(W, 0, (size_t)N * K * 4));
    CK(hipMalloc(&Wp, gemm_x3_packed_elems(N, K) * 2)); CK(launch_pack_x3(W, N, K, Wp, 0));
    CK(hipFuncSetAttribute((const void*)spin, hipFuncAttributeMaxDynamicSharedMemorySize, 52224));
    CK(hipDeviceSynchronize());
    for (int r = 0; r < reps; ++r) {
        for (int i = 0; i < NS; ++i) {
            const float tag = (float)(r * NS + i + 1);
            hipLaunchKernelGGL(produce, dim3(64 + 32 * i), dim3(256), 0, st[i], u[i], n, (n + (64 + 32 * i) * 256 - 1) / ((64 + 32 * i) * 256), tag, 8);
            if (i & 1) {
                if (use_spin) hipLaunchKernelGGL(spin, dim3(212), dim3(256), 52224, st[i], C, 4000);
                else {
                    GemmParams p{}; p.A = A; p.W = W; p.Wsplit = Wp; p.C = C + (size_t)i * M * N; p.M = M; p.N = N; p.K = K; p.lda = K; p.ldc = N;
                    CK(launch_gemm_x3(p, st[i]));
                }
            }
            hipLaunchKernelGGL(consume, dim3(256), dim3(256), 0, st[i], u[i], n, tag, bad[i]);
        }
        if ((r & 15) == 15) CK(hipDeviceSynchronize());
    }
    CK(hipDeviceSynchronize());
    unsigned long long total = 0;
    for (int i = 0; i < NS; ++i) {
        unsigned long long b; CK(hipMemcpy(&b, bad[i], 8, hipMemcpyDeviceToHost));
        printf("stream %d: %llu stale elements\n", i, b); total += b;
    }
    printf("%d streams%s, %d repetitions, aggressor %s: %llu stale elements\n", NS, prio ? " (alternating priorities)" : "", reps, use_spin ? "spin" : "mocha_gemm_x3", total);
    return 0;
}
