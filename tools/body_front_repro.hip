// Standalone attempt to reproduce the round-2 concurrency problem of mocha_body_front (tools/experiments/README.md): the ORIGINAL
// build of that kernel (72 adjacency coefficients in LDS, ext-vector arithmetic -> v_pk_fma_f32 with op_sel on LDS-fed registers)
// runs twice on the same input on one stream and the two outputs are compared, while a second stream keeps the chip busy with
// mocha_gemm_x3 launches of the shapes the pipeline had beside it.  Any difference between the two runs is the problem.
// build: tools/build_gemm_bench.sh, then
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I mocha_sigasia2023_amd/csrc -c tools/body_front_repro.hip -o tools/bin/body_front_repro.o
//   hipcc --offload-arch=gfx950 tools/bin/body_front_repro.o mocha_sigasia2023_amd/csrc/gemm_f32.o mocha_sigasia2023_amd/csrc/gemm_x3.o -o tools/bin/body_front_repro
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "kernels.h"
#include "device_utils.h"
#define main x3_mix_probe_main
#include "x3_mix_probe.hip"          // its mix<TN, SPLIT, LOADS> kernels serve as synthetic aggressors (3..7)
#undef main
#undef CK
using namespace mocha;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float lrelu02(float x) { return x > 0.f ? x : 0.2f * x; }

// VAR 0: the original (coefficients read from LDS where they are used: broadcast ds_read_b128 + counted lgkmcnt waits, ext-vector
//        arithmetic -> v_pk_fma_f32 with op_sel on the freshly returned registers)
// VAR 1: all 72 coefficients read from LDS into registers up front (long before their use), same packed arithmetic
// VAR 2: LDS coefficients made wave-uniform scalars (v_readfirstlane) before use: packed arithmetic with SGPR operands
// VAR 3: LDS coefficients where they are used, scalar arithmetic (no packed instructions)
// VAR 4: coefficients by per-lane global loads (VGPRs fed by VMEM instead of LDS), packed arithmetic
// VAR 5: LDS coefficients read up front as 36 register PAIRS (long-lived), packed arithmetic taking either half of a pair (op_sel)
#define BODY_FRONT_BODY(VAR) { \
    __shared__ float a[72]; \
    if (threadIdx.x < 72) a[threadIdx.x] = Ab[threadIdx.x]; \
    __syncthreads(); \
    const int gid = blockIdx.x * 256 + threadIdx.x; \
    const int f = gid >> 6, c4 = (gid & 63) * 4; \
    if (f >= frames) return; \
    float cf[72]; \
    f32x2_t cp[36]; \
    if (VAR == 5) { \
        _Pragma("unroll") for (int i = 0; i < 36; ++i) { cp[i][0] = a[2 * i]; cp[i][1] = a[2 * i + 1]; } \
        _Pragma("unroll") for (int i = 0; i < 36; ++i) asm volatile("" : "+v"(cp[i])); \
    } \
    const float* av = Ab + (frames < 0 ? threadIdx.x : 0);      /* VAR 4: per-lane (vector) global loads of the coefficients */ \
    if (VAR == 1) { \
_Pragma("unroll") \
        for (int i = 0; i < 72; ++i) { cf[i] = a[i]; } \
_Pragma("unroll") \
        for (int i = 0; i < 72; ++i) asm volatile("" : "+v"(cf[i])); \
    } \
    f32x4 xv[6]; \
_Pragma("unroll") \
    for (int v = 0; v < 6; ++v) { \
        f32x4 t = *reinterpret_cast<const f32x4*>(x + ((size_t)f * 6 + v) * 256 + c4); \
        t[0] = lrelu02(t[0]); t[1] = lrelu02(t[1]); t[2] = lrelu02(t[2]); t[3] = lrelu02(t[3]); \
        xv[v] = t; \
    } \
_Pragma("unroll") \
    for (int w = 0; w < 6; ++w) \
_Pragma("unroll") \
        for (int k = 0; k < 2; ++k) { \
            f32x4 acc = {0.f, 0.f, 0.f, 0.f}; \
_Pragma("unroll") \
            for (int v = 0; v < 6; ++v) { \
                const int i = (k * 6 + v) * 6 + w; \
                float c = VAR == 1 ? cf[i] : (VAR == 4 ? av[i] : (VAR == 5 ? cp[i >> 1][i & 1] : a[i])); \
                if (VAR == 2) c = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, c))); \
                acc += xv[v] * c; \
            } \
            *reinterpret_cast<f32x4*>(out + ((size_t)f * 6 + w) * 512 + k * 256 + c4) = acc; \
        } \
}
template <int VAR> __global__ __launch_bounds__(256) void body_front_var(const float* __restrict__ x, const float* __restrict__ Ab, float* __restrict__ out, int frames) BODY_FRONT_BODY(VAR)
__global__ __launch_bounds__(256) MOCHA_NO_PACKED_F32 void body_front_scalar(const float* __restrict__ x, const float* __restrict__ Ab, float* __restrict__ out, int frames) BODY_FRONT_BODY(3)

// single-instruction aggressors (8..11): which instruction of the plane split is it?
template <int KIND>
__global__ __launch_bounds__(256) void agg_one(float* out, int iters) {
    __shared__ unsigned long long sm[256];
    float a = 1.0f + threadIdx.x * 0.01f, b = 2.0f - threadIdx.x * 0.02f;
    unsigned u = threadIdx.x * 2654435761u;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            if (KIND == 0) { unsigned r; asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); u ^= r; a += 0.5f; }
            else if (KIND == 1) { unsigned r; asm volatile("v_lshlrev_b32 %0, 16, %1" : "=v"(r) : "v"(u)); asm volatile("v_and_b32 %0, 0xffff0000, %1" : "=v"(u) : "v"(r)); asm volatile("v_sub_f32 %0, %1, %2" : "=v"(a) : "v"(a), "v"(b)); u += r + 1; }
            else if (KIND == 2) { sm[threadIdx.x] = ((unsigned long long)u << 32) | (unsigned)k; u += (unsigned)sm[(threadIdx.x + 1) & 255]; }
            else { float2 r; asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(r) : "v"(make_float2(a, b)), "v"(make_float2(b, a))); a = r.x * 0.5f; b = r.y * 0.25f + 1.f; }
        }
    }
    if (u == 12345u && a == b) out[0] = a;
}

// MFMA with one kind of VALU instruction in its shadow (12..16): which VALU instruction beside the matrix pipe is it?
template <int KIND>
__global__ __launch_bounds__(256) void agg_mfma(const s16x8_t* __restrict__ src, float* out, int iters) {
    const s16x8_t a = src[threadIdx.x], b = src[256 + threadIdx.x];
    f32x16 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float x = 1.0f + threadIdx.x * 0.01f, y = 2.0f - threadIdx.x * 0.02f;
    unsigned u = threadIdx.x * 2654435761u;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[m & 3], 0, 0, 0);
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                if (KIND == 0) { unsigned r; asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y)); u ^= r; }
                else if (KIND == 1) { unsigned r; asm volatile("v_lshlrev_b32 %0, 16, %1" : "=v"(r) : "v"(u)); u = r + 1; }
                else if (KIND == 2) { asm volatile("v_sub_f32 %0, %1, %2" : "=v"(x) : "v"(x), "v"(y)); }
                else if (KIND == 3) { asm volatile("v_mov_b32 %0, %1" : "=v"(u) : "v"(u)); }
                else if (KIND == 4) { unsigned r; asm volatile("v_and_b32 %0, 0xffff0000, %1" : "=v"(r) : "v"(u)); u = r | 1; }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = x + y;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    if (u == 12345u) s += 1.f;
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

__global__ void compare(const float* a, const float* b, size_t n, unsigned long long* bad) {
    unsigned c = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) c += a[i] != b[i];
    if (c) atomicAdd(bad, (unsigned long long)c);
}

int main(int argc, char** argv) {
    const int reps = argc > 1 ? atoi(argv[1]) : 2000;
    const int windows = argc > 2 ? atoi(argv[2]) : 300;
    const int aggressor = argc > 3 ? atoi(argv[3]) : 1;            // 0 none, 1 mocha_gemm_x3, 2 the exact-f32 GEMM kernel, 3..7 synthetic (below)
    const int var = argc > 4 ? atoi(argv[4]) : 0;
    CK(gemm_init()); CK(gemm_x3_init());
    const int frames = windows * 15;
    const size_t nx = (size_t)frames * 6 * 256, no = (size_t)frames * 6 * 512;
    std::vector<float> hx(nx), hab(72, 0.f);
    for (auto& v : hx) v = (float)rand() / RAND_MAX * 2 - 1;
    for (int k = 0; k < 2; ++k) for (int v = 0; v < 6; ++v) for (int w = 0; w < 6; ++w)          // a sparse adjacency: self loops and neighbours
        hab[(k * 6 + v) * 6 + w] = (k == 0 ? (v == w ? 0.5f : 0.f) : ((v + 1) % 6 == w || (w + 1) % 6 == v ? 0.25f : 0.f));
    const int cpat = argc > 5 ? atoi(argv[5]) : 0;             // coefficient pattern: 0 the sparse adjacency, 1 both halves of every register pair equal, 2 all distinct
    if (cpat == 1) for (int i = 0; i < 72; ++i) hab[i] = 0.125f * (1 + i / 2);
    if (cpat == 2) for (int i = 0; i < 72; ++i) hab[i] = 0.03125f * (1 + i);
    float *x, *ab, *o1, *o2; unsigned long long* bad;
    CK(hipMalloc(&x, nx * 4)); CK(hipMalloc(&ab, 72 * 4)); CK(hipMalloc(&o1, no * 4)); CK(hipMalloc(&o2, no * 4)); CK(hipMalloc(&bad, 8)); CK(hipMemset(bad, 0, 8));
    CK(hipMemcpy(x, hx.data(), nx * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(ab, hab.data(), 72 * 4, hipMemcpyHostToDevice));
    // aggressor operands: the embedding's GEMMs at this batch
    struct G { int M, N, K; float *A, *C; unsigned short* Wp; GemmParams p; } gs[3] = {{windows * 360, 256, 192}, {windows * 90, 256, 1280}, {windows * 90, 3072, 256}};
    for (auto& g : gs) {
        float* W; CK(hipMalloc(&g.A, (size_t)g.M * g.K * 4)); CK(hipMalloc(&W, (size_t)g.N * g.K * 4)); CK(hipMalloc(&g.C, (size_t)g.M * g.N * 4));
        CK(hipMemset(g.A, 0x3c, (size_t)g.M * g.K * 4)); CK(hipMemset(W, 0x3c, (size_t)g.N * g.K * 4));
        CK(hipMalloc(&g.Wp, gemm_x3_packed_elems(g.N, g.K) * 2)); CK(launch_pack_x3(W, g.N, g.K, g.Wp, 0));
        g.p = GemmParams{}; g.p.A = g.A; g.p.W = W; g.p.Wsplit = g.Wp; g.p.C = g.C; g.p.M = g.M; g.p.N = g.N; g.p.K = g.K; g.p.lda = g.K; g.p.ldc = g.N;
    }
    // synthetic aggressors: the pieces of a mocha_gemm_x3 K step (tools/x3_mix_probe.hip)
    s16x8_t* md; float *mo, *mact; unsigned short* mw;
    {
        std::vector<unsigned short> hm(12 * 256 * 8);
        for (auto& v : hm) { const unsigned r = (unsigned)rand(); v = (unsigned short)(((r & 1) << 15) | ((126 + ((r >> 1) & 1)) << 7) | ((r >> 2) & 0x7f)); }
        CK(hipMalloc(&md, hm.size() * 2)); CK(hipMemcpy(md, hm.data(), hm.size() * 2, hipMemcpyHostToDevice));
        CK(hipMalloc(&mo, (size_t)4096 * 256 * 4)); CK(hipMalloc(&mact, (size_t)800 * 128 * 256 * 4)); CK(hipMemset(mact, 0, (size_t)800 * 128 * 256 * 4));
        CK(hipMalloc(&mw, (size_t)4 * 16 * 6144 * 2 * 2)); CK(hipMemset(mw, 0, (size_t)4 * 16 * 6144 * 2 * 2));
    }
    const size_t mlds = (size_t)6 * 256 * 16 + 6 * 256 * 8 + (size_t)6144 * 2 + 2048;
    hipStream_t sa, sb; CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
    CK(hipDeviceSynchronize());
    const dim3 grid((unsigned)(((long long)frames * 64 + 255) / 256));
    for (int r = 0; r < reps; ++r) {
        if (aggressor == 1) CK(launch_gemm_x3(gs[r % 3].p, sa));
        else if (aggressor == 2) CK(launch_gemm(gs[r % 3].p, sa));
        else if (aggressor == 3) hipLaunchKernelGGL((mix<2, false, 0>), dim3(768), dim3(256), mlds, sa, md, mo, 100, mact, mw);
        else if (aggressor == 4) hipLaunchKernelGGL((mix<2, true, 0>), dim3(768), dim3(256), mlds, sa, md, mo, 100, mact, mw);
        else if (aggressor == 5) hipLaunchKernelGGL((mix<2, true, 1>), dim3(768), dim3(256), mlds, sa, md, mo, 100, mact, mw);
        else if (aggressor == 6) hipLaunchKernelGGL((mix<2, true, 2>), dim3(768), dim3(256), mlds, sa, md, mo, 100, mact, mw);
        else if (aggressor == 7) hipLaunchKernelGGL((mix<2, true, 3>), dim3(768), dim3(256), mlds, sa, md, mo, 100, mact, mw);
        else if (aggressor == 8) hipLaunchKernelGGL(agg_one<0>, dim3(1024), dim3(256), 0, sa, mo, 300);
        else if (aggressor == 9) hipLaunchKernelGGL(agg_one<1>, dim3(1024), dim3(256), 0, sa, mo, 300);
        else if (aggressor == 10) hipLaunchKernelGGL(agg_one<2>, dim3(1024), dim3(256), 0, sa, mo, 300);
        else if (aggressor == 11) hipLaunchKernelGGL(agg_one<3>, dim3(1024), dim3(256), 0, sa, mo, 300);
        else if (aggressor == 12) hipLaunchKernelGGL(agg_mfma<0>, dim3(1024), dim3(256), 0, sa, md, mo, 300);
        else if (aggressor == 13) hipLaunchKernelGGL(agg_mfma<1>, dim3(1024), dim3(256), 0, sa, md, mo, 300);
        else if (aggressor == 14) hipLaunchKernelGGL(agg_mfma<2>, dim3(1024), dim3(256), 0, sa, md, mo, 300);
        else if (aggressor == 15) hipLaunchKernelGGL(agg_mfma<3>, dim3(1024), dim3(256), 0, sa, md, mo, 300);
        else if (aggressor == 16) hipLaunchKernelGGL(agg_mfma<4>, dim3(1024), dim3(256), 0, sa, md, mo, 300);
        for (float* o : {o1, o2}) {
            if (var == 0) hipLaunchKernelGGL(body_front_var<0>, grid, dim3(256), 0, sb, x, ab, o, frames);
            else if (var == 1) hipLaunchKernelGGL(body_front_var<1>, grid, dim3(256), 0, sb, x, ab, o, frames);
            else if (var == 2) hipLaunchKernelGGL(body_front_var<2>, grid, dim3(256), 0, sb, x, ab, o, frames);
            else if (var == 4) hipLaunchKernelGGL(body_front_var<4>, grid, dim3(256), 0, sb, x, ab, o, frames);
            else if (var == 5) hipLaunchKernelGGL(body_front_var<5>, grid, dim3(256), 0, sb, x, ab, o, frames);
            else hipLaunchKernelGGL(body_front_scalar, grid, dim3(256), 0, sb, x, ab, o, frames);
        }
        hipLaunchKernelGGL(compare, dim3(512), dim3(256), 0, sb, o1, o2, no, bad);
        if ((r & 31) == 31) CK(hipDeviceSynchronize());
    }
    CK(hipDeviceSynchronize());
    unsigned long long hb; CK(hipMemcpy(&hb, bad, 8, hipMemcpyDeviceToHost));
    const char* an[17] = {"nothing beside it", "mocha_gemm_x3 on a second stream", "mocha_gemm_f32 on a second stream", "bf16 MFMA + LDS reads", "... + plane split (cvt_pk, ds_write)", "... + split + buffer loads", "... + split + LDS-DMA", "... + split + loads + LDS-DMA", "v_cvt_pk_bf16_f32 only", "v_lshlrev / v_and / v_sub_f32 only", "ds_write_b64 + ds_read only", "v_pk_mul_f32 only", "bf16 MFMA + v_cvt_pk_bf16_f32", "bf16 MFMA + v_lshlrev_b32", "bf16 MFMA + v_sub_f32", "bf16 MFMA + v_mov_b32", "bf16 MFMA + v_and_b32 (literal)"};
    const char* vn[6] = {"LDS coefficients at use + packed fmas (the original)", "LDS coefficients read up front + packed fmas", "LDS coefficients via readfirstlane + packed fmas", "LDS coefficients at use + scalar fmas", "coefficients by vector global loads + packed fmas", "LDS coefficient PAIRS read up front + packed fmas"};
    printf("%-52s | %-34s | %d x %d windows: %llu elements differ between two runs on the same input\n", vn[var], an[aggressor], reps, windows, hb);
    if (cpat) printf("   (coefficient pattern %d: %s)\n", cpat, cpat == 1 ? "both halves of every coefficient pair equal" : "all coefficients distinct");
    return 0;
}
