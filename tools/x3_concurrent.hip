// Two mocha_gemm_x3 launches at a time on two streams (mid-size shapes that do not fill the chip, so that their workgroups share
// CUs), each checked against a float64 product.  build: see tools/build_gemm_bench.sh (links gemm_f32.o gemm_x3.o)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include "kernels.h"
using namespace mocha;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

__global__ void ref_gemm(const float* A, const float* W, double* C, int M, int N, int K, const float* bias, const float* res) {
    int n = blockIdx.x * 16 + threadIdx.x, m = blockIdx.y * 16 + threadIdx.y;
    if (m >= M || n >= N) return;
    double a = 0;
    for (int k = 0; k < K; ++k) a += (double)A[(size_t)m * K + k] * W[(size_t)n * K + k];
    if (bias) a += bias[(size_t)(m % 6) * N + n];       // used as a row bias with period 6
    if (res) a += res[(size_t)m * N + n];
    C[(size_t)m * N + n] = a;
}

struct Job { int M, N, K; float *A, *W, *C, *B, *Rs; unsigned short* Wp; double* R; hipStream_t s; GemmParams p; };

int main(int argc, char** argv) {
    CK(gemm_init()); CK(gemm_x3_init());
    const int reps = argc > 1 ? atoi(argv[1]) : 50;
    Job jobs[2] = {{6750, 256, 512}, {6750, 512, 256}};
    if (argc > 4) { jobs[0].M = jobs[1].M = atoi(argv[2]); jobs[0].N = atoi(argv[3]); jobs[1].N = atoi(argv[4]); }
    for (auto& j : jobs) {
        std::vector<float> ha((size_t)j.M * j.K), hw((size_t)j.N * j.K);
        for (auto& v : ha) v = (float)rand() / RAND_MAX * 2 - 1;
        for (auto& v : hw) v = (float)rand() / RAND_MAX * 2 - 1;
        CK(hipMalloc(&j.A, ha.size() * 4)); CK(hipMalloc(&j.W, hw.size() * 4)); CK(hipMalloc(&j.C, (size_t)j.M * j.N * 4)); CK(hipMalloc(&j.R, (size_t)j.M * j.N * 8));
        CK(hipMalloc(&j.Wp, gemm_x3_packed_elems(j.N, j.K) * 2));
        std::vector<float> hb((size_t)6 * j.N), hres((size_t)j.M * j.N);
        for (auto& v : hb) v = (float)rand() / RAND_MAX; for (auto& v : hres) v = (float)rand() / RAND_MAX;
        CK(hipMalloc(&j.B, hb.size() * 4)); CK(hipMalloc(&j.Rs, hres.size() * 4));
        CK(hipMemcpy(j.B, hb.data(), hb.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(j.Rs, hres.data(), hres.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(j.A, ha.data(), ha.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(j.W, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
        CK(hipStreamCreateWithFlags(&j.s, hipStreamNonBlocking));
        CK(launch_pack_x3(j.W, j.N, j.K, j.Wp, 0));
        hipLaunchKernelGGL(ref_gemm, dim3((j.N + 15) / 16, (j.M + 15) / 16), dim3(16, 16), 0, 0, j.A, j.W, j.R, j.M, j.N, j.K, j.B, getenv("X3C_RES") ? j.Rs : nullptr);
        j.p.A = j.A; j.p.W = j.W; j.p.Wsplit = j.Wp; j.p.C = j.C; j.p.M = j.M; j.p.N = j.N; j.p.K = j.K; j.p.lda = j.K; j.p.ldc = j.N; j.p.rowbias = j.B; j.p.rb_mod = 6; if (getenv("X3C_RES")) { j.p.residual = j.Rs; j.p.ldr = j.N; }
    }
    CK(hipDeviceSynchronize());
    for (int mode = 0; mode < 2; ++mode) {               // 0: one after the other on one stream, 1: concurrently on two
        long bad_total = 0;
        for (int r = 0; r < reps; ++r) {
            for (auto& j : jobs) { CK(hipMemsetAsync(j.C, 0xff, (size_t)j.M * j.N * 4, mode ? j.s : 0)); }
            for (auto& j : jobs) CK(launch_gemm_x3(j.p, mode ? j.s : 0));
            CK(hipDeviceSynchronize());
            for (auto& j : jobs) {
                std::vector<float> hc((size_t)j.M * j.N); std::vector<double> hr(hc.size());
                CK(hipMemcpy(hc.data(), j.C, hc.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(hr.data(), j.R, hr.size() * 8, hipMemcpyDeviceToHost));
                long bad = 0; long first = -1;
                for (size_t i = 0; i < hc.size(); ++i) if (!(fabs((double)hc[i] - hr[i]) < 1e-3)) { ++bad; if (first < 0) first = (long)i; }
                if (bad) printf("  mode %d rep %d job N=%d: %ld wrong elements, first at (%ld, %ld)\n", mode, r, j.N, bad, first / j.N, first % j.N);
                bad_total += bad;
            }
        }
        printf("%s: %ld wrong elements over %d repetitions\n", mode ? "two streams" : "one stream", bad_total, reps);
    }
    return 0;
}
