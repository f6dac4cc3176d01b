// What would a 128 x 256 tile buy mocha_gemm_x3?  The instruction mix of one K step, without the memory side, at the two shapes:
//   TN = 2 (shipped, 128 x 128 tile, wave = 64 x 64):  24 MFMAs, 12 ds_read_b128, 44 split VALU, 6 ds_write_b64, 1 barrier; 3 WGs / CU
//   TN = 1 (128 x 64 tile, wave = 64 x 32, shipped for N = 64 / 192): 12 MFMAs, 9 ds_read_b128, 44 split VALU, 6 ds_write_b64; 3 WGs / CU
//   TN = 4 (128 x 256 tile, wave = 64 x 128):          48 MFMAs, 18 ds_read_b128, 44 split VALU, 6 ds_write_b64, 1 barrier; 2 WGs / CU
// optionally with the memory side of the step (two 16-byte activation loads per thread, the 12 KB x TN / 2 weight copy by LDS-DMA),
// launched back to back for a few seconds (the board settles at the clock it holds under that load).  Operands are random bf16.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I mocha_sigasia2023_amd/csrc tools/x3_mix_probe.hip -o tools/bin/x3_mix_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "device_utils.h"
using namespace mocha;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int TN, bool SPLIT, int LOADS>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(TN == 4 ? 2 : 3, TN == 4 ? 2 : 3)))
void mix(const s16x8_t* __restrict__ src, float* __restrict__ out, int iters, const float* __restrict__ act, const unsigned short* __restrict__ wimg) {
    extern __shared__ __attribute__((aligned(16))) s16x8_t sm[];      // operands: 6 x 256 fragments, then 6 x 256 x 8 B of plane writes, then the copied weights
    constexpr int NFRAG = 6;                          // six 4 KB fragment slots shared by the A and B reads (the counts are what matters)
    const int tid = threadIdx.x;
    for (int i = tid; i < NFRAG * 256; i += 256) sm[i] = src[i % (12 * 256)];
    u32x2_t* wr = reinterpret_cast<u32x2_t*>(sm + NFRAG * 256);
    __syncthreads();
    f32x16 acc[2][TN];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    float xs[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) xs[e] = 0.37f * (tid + 1) + e;
    // the memory side of a K step (LOADS bit 0: the activation fetch, two 16-byte loads per thread from this workgroup's 128 rows of
    // a (rows x 256) fp32 matrix, 64 bytes per row and step; bit 1: the weight copy, 12 KB x TN / 2 per step by LDS-DMA from an
    // L2-resident image)
    const __amdgpu_buffer_rsrc_t rsA = make_rsrc(act + (size_t)(blockIdx.x % 800) * 128 * 256);
    const __amdgpu_buffer_rsrc_t rsW = make_rsrc(wimg + (size_t)(blockIdx.x & 1) * 16 * 3072 * TN);
    const unsigned a_off = ((unsigned)(tid >> 2) * 256u + (tid & 3) * 4u) * 4u;
    unsigned short* dma_dst = reinterpret_cast<unsigned short*>(sm + NFRAG * 256) + 6 * 256 * 4;      // after the plane writes
    f32x4_t fa[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
    for (int it = 0; it < iters; ++it) {
        if (LOADS & 2) {
#pragma unroll
            for (int j = 0; j < (3 * TN + 1) / 2; ++j)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (__attribute__((address_space(3))) void*)(dma_dst + (j * 4 + (tid >> 6)) * 512), 16,
                                                         (unsigned)(j * 256 + tid) * 16u, (unsigned)(it & 15) * (6144u * TN), 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (LOADS & 1) {
            xs[0] += fa[0][0] + fa[1][1];              // consume the previous step's fetch
            fa[0] = bload(rsA, a_off, (unsigned)(it & 15) * 64u);
            fa[1] = bload(rsA, a_off + 64u * 256u * 4u, (unsigned)(it & 15) * 64u);
        }
        __builtin_amdgcn_sched_barrier(0);
        s16x8_t a[3][2], b[3][TN];
#pragma unroll
        for (int q = 0; q < 3; ++q) {
#pragma unroll
            for (int i = 0; i < 2; ++i) a[q][i] = sm[(q * 2 + i) * 256 + ((tid + it) & 255)];
#pragma unroll
            for (int j = 0; j < TN; ++j) b[q][j] = sm[((q * TN + j) % 6) * 256 + ((tid + 7 * it + 64 * ((q * TN + j) / 6)) & 255)];
        }
        float x[8];
        unsigned pk[4][3], hi[4][2];
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] = xs[e];
        auto split_op = [&](int k) __attribute__((always_inline)) {
            const int pr = k / 11, o = k % 11, lvl = o / 5;
            float& x0 = x[2 * pr]; float& x1 = x[2 * pr + 1];
            if (o == 10) { pk[pr][2] = cvt_pk_bf16(x0, x1); return; }
            switch (o % 5) {
                case 0: pk[pr][lvl] = cvt_pk_bf16(x0, x1); break;
                case 1: hi[pr][0] = pk[pr][lvl] << 16; break;
                case 2: hi[pr][1] = pk[pr][lvl] & 0xffff0000u; break;
                case 3: x0 -= __uint_as_float(hi[pr][0]); break;
                default: x1 -= __uint_as_float(hi[pr][1]); break;
            }
        };
        auto write_row = [&](int i) __attribute__((always_inline)) {
#pragma unroll
            for (int q = 0; q < 3; ++q) { const u32x2_t v = {pk[2 * i][q], pk[2 * i + 1][q]}; wr[(q * 2 + i) * 256 + tid] = v; }
        };
        constexpr int NM = 12 * TN;                      // MFMAs per step
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = 0; m < NM; ++m) {
            const int pr = m / (2 * TN), r = m % (2 * TN), i = r / TN, j = r % TN;
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[PLANE_PB[pr]][j], a[PLANE_PA[pr]][i], acc[i][j], 0, 0, 0);
            if (SPLIT) {
                if (TN == 1) { if (m < 11) { split_op(4 * m); split_op(4 * m + 1); split_op(4 * m + 2); split_op(4 * m + 3); } if (m == 5) write_row(0); if (m == 10) write_row(1); }
                else if (TN == 2) { if (m < 22) { split_op(2 * m); split_op(2 * m + 1); } if (m == 10) write_row(0); if (m == 21) write_row(1); }
                else { if (m < 44) split_op(m); if (m == 21) write_row(0); if (m == 43) write_row(1); }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (LOADS & 1) asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll
        for (int e = 0; e < 8; ++e) xs[e] += 0.001f;
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    out[blockIdx.x * 256 + tid] = s + xs[0] + fa[0][2] + fa[1][3];
}

// The same K step of the 128 x 128 tile on v_mfma_f32_16x16x32_bf16 (the hardware guide reports that shape to hold a higher clock under
// the power cap): a wave's 64 x 64 tile is 16 blocks of 16 x 16; one instruction contracts K = 32, used here as TWO plane products of
// the same 16 k (operand A = [plane x | plane y] along K, operand B = [plane y' | plane x']), so the six products of a step are three
// instructions per block: 48 MFMAs, and 24 fragment reads (4 row blocks x 3 plane pairs for A, the same for B) instead of 12.
typedef float f32x4v __attribute__((ext_vector_type(4)));
template <bool SPLIT, int LOADS>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3)))
void mix16(const s16x8_t* __restrict__ src, float* __restrict__ out, int iters, const float* __restrict__ act, const unsigned short* __restrict__ wimg) {
    extern __shared__ __attribute__((aligned(16))) s16x8_t sm[];
    constexpr int NFRAG = 6;
    const int tid = threadIdx.x;
    for (int i = tid; i < NFRAG * 256; i += 256) sm[i] = src[i % (12 * 256)];
    u32x2_t* wr = reinterpret_cast<u32x2_t*>(sm + NFRAG * 256);
    __syncthreads();
    f32x4v acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4v{0.f, 0.f, 0.f, 0.f};
    float xs[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) xs[e] = 0.37f * (tid + 1) + e;
    const __amdgpu_buffer_rsrc_t rsA = make_rsrc(act + (size_t)(blockIdx.x % 800) * 128 * 256);
    const __amdgpu_buffer_rsrc_t rsW = make_rsrc(wimg + (size_t)(blockIdx.x & 1) * 16 * 3072 * 2);
    const unsigned a_off = ((unsigned)(tid >> 2) * 256u + (tid & 3) * 4u) * 4u;
    unsigned short* dma_dst = reinterpret_cast<unsigned short*>(sm + NFRAG * 256) + 6 * 256 * 4;
    f32x4_t fa[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
    for (int it = 0; it < iters; ++it) {
        if (LOADS & 2) {
#pragma unroll
            for (int j = 0; j < 3; ++j)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (__attribute__((address_space(3))) void*)(dma_dst + (j * 4 + (tid >> 6)) * 512), 16,
                                                         (unsigned)(j * 256 + tid) * 16u, (unsigned)(it & 15) * (6144u * 2), 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (LOADS & 1) {
            xs[0] += fa[0][0] + fa[1][1];
            fa[0] = bload(rsA, a_off, (unsigned)(it & 15) * 64u);
            fa[1] = bload(rsA, a_off + 64u * 256u * 4u, (unsigned)(it & 15) * 64u);
        }
        __builtin_amdgcn_sched_barrier(0);
        s16x8_t a[3][4], b[3][4];                          // [plane pair][row / column block]
#pragma unroll
        for (int q = 0; q < 3; ++q)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                a[q][i] = sm[((q * 4 + i) % 6) * 256 + ((tid + it + 32 * ((q * 4 + i) / 6)) & 255)];
                b[q][i] = sm[((q * 4 + i + 3) % 6) * 256 + ((tid + 7 * it + 64 * ((q * 4 + i) / 6)) & 255)];
            }
        float x[8];
        unsigned pk[4][3], hi[4][2];
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] = xs[e];
        auto split_op = [&](int k) __attribute__((always_inline)) {
            const int pr = k / 11, o = k % 11, lvl = o / 5;
            float& x0 = x[2 * pr]; float& x1 = x[2 * pr + 1];
            if (o == 10) { pk[pr][2] = cvt_pk_bf16(x0, x1); return; }
            switch (o % 5) {
                case 0: pk[pr][lvl] = cvt_pk_bf16(x0, x1); break;
                case 1: hi[pr][0] = pk[pr][lvl] << 16; break;
                case 2: hi[pr][1] = pk[pr][lvl] & 0xffff0000u; break;
                case 3: x0 -= __uint_as_float(hi[pr][0]); break;
                default: x1 -= __uint_as_float(hi[pr][1]); break;
            }
        };
        auto write_row = [&](int i) __attribute__((always_inline)) {
#pragma unroll
            for (int q = 0; q < 3; ++q) { const u32x2_t v = {pk[2 * i][q], pk[2 * i + 1][q]}; wr[(q * 2 + i) * 256 + tid] = v; }
        };
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = 0; m < 48; ++m) {
            const int q = m / 16, i = (m % 16) / 4, j = m % 4;
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[q][j], a[q][i], acc[i][j], 0, 0, 0);
            if (SPLIT) { if (m < 44) split_op(m); if (m == 21) write_row(0); if (m == 43) write_row(1); }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (LOADS & 1) asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll
        for (int e = 0; e < 8; ++e) xs[e] += 0.001f;
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    out[blockIdx.x * 256 + tid] = s + xs[0] + fa[0][2] + fa[1][3];
}

template <bool SPLIT, int LOADS>
static void run16(const s16x8_t* d, float* o, int iters, double seconds, const char* name, const float* act, const unsigned short* wimg) {
    const size_t lds = (size_t)6 * 256 * 16 + 6 * 256 * 8 + (size_t)6144 * 2 + 2048;
    CK(hipFuncSetAttribute((const void*)mix16<SPLIT, LOADS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int wgs = 256 * 3 * 2;
    auto launch = [&]() { hipLaunchKernelGGL((mix16<SPLIT, LOADS>), dim3(wgs), dim3(256), lds, 0, d, o, iters, act, wimg); };
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    launch(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0)); launch(); CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms1; CK(hipEventElapsedTime(&ms1, e0, e1));
    const int n = (int)(seconds * 1e3 / ms1) + 1;
    for (int i = 0; i < n; ++i) launch();
    const int tail = n / 10 + 1;
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < tail; ++i) launch();
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double flops = (double)wgs * 4 * iters * 48 * (2.0 * 16 * 16 * 32);
    printf("%-72s first %7.1f   settled %7.1f TFLOP/s = %.2f of 2516.8   (LDS %zu B)\n", name, flops / (ms1 * 1e-3) / 1e12, flops * tail / (ms * 1e-3) / 1e12,
           flops * tail / (ms * 1e-3) / 1e12 / 2516.8, lds);
}

template <int TN, bool SPLIT, int LOADS>
static void run(const s16x8_t* d, float* o, int iters, double seconds, const char* name, const float* act, const unsigned short* wimg) {
    const size_t lds = (size_t)6 * 256 * 16 + 6 * 256 * 8 + (size_t)6144 * TN + 2048;
    CK(hipFuncSetAttribute((const void*)mix<TN, SPLIT, LOADS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int wgs = 256 * (TN == 4 ? 2 : 3) * 2;
    auto launch = [&]() { hipLaunchKernelGGL((mix<TN, SPLIT, LOADS>), dim3(wgs), dim3(256), lds, 0, d, o, iters, act, wimg); };
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    launch(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0)); launch(); CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms1; CK(hipEventElapsedTime(&ms1, e0, e1));
    const int n = (int)(seconds * 1e3 / ms1) + 1;
    for (int i = 0; i < n; ++i) launch();
    const int tail = n / 10 + 1;
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < tail; ++i) launch();
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double flops = (double)wgs * 4 * iters * (12 * TN) * (2.0 * 32 * 32 * 16);
    printf("%-72s first %7.1f   settled %7.1f TFLOP/s = %.2f of 2516.8   (LDS %zu B)\n", name, flops / (ms1 * 1e-3) / 1e12, flops * tail / (ms * 1e-3) / 1e12,
           flops * tail / (ms * 1e-3) / 1e12 / 2516.8, lds);
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 1000;
    const double seconds = argc > 2 ? atof(argv[2]) : 2.0;
    std::vector<unsigned short> h(12 * 256 * 8);
    for (auto& v : h) { const unsigned r = (unsigned)rand(); v = (unsigned short)(((r & 1) << 15) | ((126 + ((r >> 1) & 1)) << 7) | ((r >> 2) & 0x7f)); }
    s16x8_t* d; float* o;
    CK(hipMalloc(&d, h.size() * 2)); CK(hipMalloc(&o, (size_t)4096 * 256 * 4));
    CK(hipMemcpy(d, h.data(), h.size() * 2, hipMemcpyHostToDevice));
    float* act; unsigned short* wimg;
    CK(hipMalloc(&act, (size_t)800 * 128 * 256 * 4)); CK(hipMemset(act, 0, (size_t)800 * 128 * 256 * 4));       // 105 MB of activations (zeros: they only feed xs)
    CK(hipMalloc(&wimg, (size_t)4 * 16 * 6144 * 2 * 2)); CK(hipMemset(wimg, 0, (size_t)4 * 16 * 6144 * 2 * 2));
    run<2, false, 0>(d, o, iters, seconds, "128 x 128: 24 MFMA + 12 reads + barrier", act, wimg);
    run<2, true, 0>(d, o, iters, seconds, "128 x 128: + 44 split VALU + 6 plane writes", act, wimg);
    run<2, true, 1>(d, o, iters, seconds, "128 x 128: + split + activation fetch (8 KB / step)", act, wimg);
    run<2, true, 2>(d, o, iters, seconds, "128 x 128: + split + weight copy by LDS-DMA (12 KB / step)", act, wimg);
    run<2, true, 3>(d, o, iters, seconds, "128 x 128: + split + both (a K step of mocha_gemm_x3)", act, wimg);
    run16<false, 0>(d, o, iters, seconds, "128 x 128 on 16x16x32: 48 MFMA + 24 reads + barrier", act, wimg);
    run16<true, 0>(d, o, iters, seconds, "128 x 128 on 16x16x32: + 44 split VALU + 6 plane writes", act, wimg);
    run16<true, 3>(d, o, iters, seconds, "128 x 128 on 16x16x32: + split + both (what a K step would be)", act, wimg);
    run<2, true, 3>(d, o, iters, seconds, "128 x 128: + split + both, again (32x32x16, after the 16x16x32 runs)", act, wimg);
    run<1, true, 3>(d, o, iters * 2, seconds, "128 x  64: 12 MFMA + 9 reads + split + fetch (8 KB) + weight copy (8 KB)", act, wimg);
    run<4, false, 0>(d, o, iters / 2, seconds, "128 x 256: 48 MFMA + 18 reads + barrier", act, wimg);
    run<4, true, 0>(d, o, iters / 2, seconds, "128 x 256: + 44 split VALU + 6 plane writes", act, wimg);
    run<4, true, 3>(d, o, iters / 2, seconds, "128 x 256: + split + fetch (8 KB) + weight copy (24 KB)", act, wimg);
    return 0;
}
