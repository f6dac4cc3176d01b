// Does a packed fp32 instruction whose LOW result takes the HIGH half of a source (v_pk_fma_f32 ... op_sel:[0,1,0]) survive a
// matrix-pipe kernel of another stream running on the same SIMDs?  Background: tools/experiments/README.md, "two streams".
// In the pipeline, mocha_body_front (the only hot-path kernel the compiler had given such instructions) intermittently stored 0 in
// the low halves of those results, lanes 48-63, when a second stream's plane-GEMM workgroups shared its CUs.
//
// victim: every lane repeats r = pk_fma(a, b, c) in four flavours and compares with two scalar v_fma_f32:
//     0: no op_sel                      r.lo = a.lo * b.lo + c.lo      r.hi = a.hi * b.hi + c.hi
//     1: op_sel_hi:[1,0,1]              r.lo = a.lo * b.lo + c.lo      r.hi = a.hi * b.lo + c.hi     (broadcast of the low half)
//     2: op_sel:[0,1,0] op_sel_hi:[1,1,1]   r.lo = a.lo * b.hi + c.lo  r.hi = a.hi * b.hi + c.hi     (broadcast of the high half)
//     3: op_sel:[0,1,0] op_sel_hi:[1,0,1]   r.lo = a.lo * b.hi + c.lo  r.hi = a.hi * b.lo + c.hi     (swap)
// and counts wrong results per (flavour, half, quarter of the wave).
// aggressor: a bf16 MFMA loop with LDS operand reads (the inner loop of tools/bf16_roof.hip), one workgroup per CU, on its own stream.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/pk_opsel_probe.hip -o tools/bin/pk_opsel_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

__global__ __launch_bounds__(256) void victim(unsigned long long* bad /*[8][2][4]*/, int iters, float seed) {
    const int lane = threadIdx.x & 63;
#ifdef PROBE_BIG_ALLOC
    asm volatile("v_mov_b32 v125, 0" ::: "v125");       // a 128-register allocation per wave, as mocha_body_front had: waves sit all over the 512-entry file
#endif
    f32x2 a = {seed + 0.5f * lane, 1.25f + 0.25f * lane}, b = {2.0f + lane, -3.0f - 0.5f * lane}, c = {0.125f * lane, 7.0f};
    unsigned wrong[8][2] = {};
    for (int it = 0; it < iters; ++it) {
        f32x2 r0, r1, r2, r3, r4, r5, r6, r7, t6, t7;
        asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(r0) : "v"(a), "v"(b), "v"(c));
        asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1]" : "=v"(r1) : "v"(a), "v"(b), "v"(c));
        asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "=v"(r2) : "v"(a), "v"(b), "v"(c));
        asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,0,1]" : "=v"(r3) : "v"(a), "v"(b), "v"(c));
        // the two forms mocha_body_front's first term of a sum had: inline constant 0 as the addend
        asm volatile("v_pk_fma_f32 %0, %1, %2, 0 op_sel_hi:[1,0,0]" : "=v"(r4) : "v"(a), "v"(b));                     // a * b.lo
        asm volatile("v_pk_fma_f32 %0, %1, %2, 0 op_sel:[0,1,0] op_sel_hi:[1,1,0]" : "=v"(r5) : "v"(a), "v"(b));      // a * b.hi
        float m_ll, m_lh, m_hl, m_hh;
        asm volatile("v_mul_f32 %0, %1, %2" : "=v"(m_ll) : "v"(a[0]), "v"(b[0]));
        asm volatile("v_mul_f32 %0, %1, %2" : "=v"(m_lh) : "v"(a[0]), "v"(b[1]));
        asm volatile("v_mul_f32 %0, %1, %2" : "=v"(m_hl) : "v"(a[1]), "v"(b[0]));
        asm volatile("v_mul_f32 %0, %1, %2" : "=v"(m_hh) : "v"(a[1]), "v"(b[1]));
        wrong[4][0] += r4[0] != m_ll; wrong[4][1] += r4[1] != m_hl;
        wrong[5][0] += r5[0] != m_lh; wrong[5][1] += r5[1] != m_hh;
        // dependent pairs, back to back, as in a sum of products: the first result is the next instruction's addend
        asm volatile("v_pk_fma_f32 %0, %2, %3, 0 op_sel:[0,1,0] op_sel_hi:[1,1,0]\n\tv_pk_fma_f32 %1, %4, %3, %0 op_sel_hi:[1,0,1]"
                     : "=&v"(t6), "=&v"(r6) : "v"(a), "v"(b), "v"(c));                       // r6 = c * b.lo + (a * b.hi)
        asm volatile("v_pk_fma_f32 %0, %2, %3, 0 op_sel_hi:[1,0,0]\n\tv_pk_fma_f32 %1, %4, %3, %0 op_sel:[0,1,0]"
                     : "=&v"(t7), "=&v"(r7) : "v"(a), "v"(b), "v"(c));                       // r7 = c * b.hi + (a * b.lo)
        {
            float p0, p1, q0, q1;
            asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(p0) : "v"(c[0]), "v"(b[0]), "v"(m_lh));
            asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(p1) : "v"(c[1]), "v"(b[0]), "v"(m_hh));
            asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(q0) : "v"(c[0]), "v"(b[1]), "v"(m_ll));
            asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(q1) : "v"(c[1]), "v"(b[1]), "v"(m_hl));
            wrong[6][0] += r6[0] != p0; wrong[6][1] += r6[1] != p1;
            wrong[7][0] += r7[0] != q0; wrong[7][1] += r7[1] != q1;
        }
        float ll, lh, hl, hh;                       // a.lo*b.lo+c.lo, a.lo*b.hi+c.lo, a.hi*b.lo+c.hi, a.hi*b.hi+c.hi
        asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(ll) : "v"(a[0]), "v"(b[0]), "v"(c[0]));
        asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(lh) : "v"(a[0]), "v"(b[1]), "v"(c[0]));
        asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(hl) : "v"(a[1]), "v"(b[0]), "v"(c[1]));
        asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(hh) : "v"(a[1]), "v"(b[1]), "v"(c[1]));
        wrong[0][0] += r0[0] != ll; wrong[0][1] += r0[1] != hh;
        wrong[1][0] += r1[0] != ll; wrong[1][1] += r1[1] != hl;
        wrong[2][0] += r2[0] != lh; wrong[2][1] += r2[1] != hh;
        wrong[3][0] += r3[0] != lh; wrong[3][1] += r3[1] != hl;
        a[0] += 1.0f; b[1] -= 0.5f; c[0] += 0.25f;          // new operands every round
    }
#pragma unroll
    for (int f = 0; f < 8; ++f)
#pragma unroll
        for (int h = 0; h < 2; ++h)
            if (wrong[f][h]) atomicAdd(bad + (f * 2 + h) * 4 + (lane >> 4), (unsigned long long)wrong[f][h]);
}

// the pipeline's pattern: the coefficient pair comes straight from a broadcast ds_read_b128 (four coefficients, every lane the same
// address), s_waitcnt lgkmcnt(0), then the packed fma takes its HIGH register for the low result
__global__ __launch_bounds__(256) void victim_lds(unsigned long long* bad /*[4][2][4] (flavours 2 and 3 used)*/, int iters, float seed) {
    __shared__ __attribute__((aligned(16))) float cf[64];
    const int lane = threadIdx.x & 63;
    if (threadIdx.x < 64) cf[threadIdx.x] = seed + 1.0f + 0.5f * threadIdx.x;
    __syncthreads();
    f32x2 a = {seed + 0.5f * lane, 1.25f + 0.25f * lane}, c = {0.125f * lane, 7.0f};
    unsigned wrong[2][2] = {};
    for (int it = 0; it < iters; ++it) {
        const unsigned addr = (unsigned)(size_t)cf + (unsigned)(it & 15) * 16u;
        f32x2 r2, r3;
        float b0, b1;
        // fixed registers for the LDS destination, so that the packed fma can name the first pair of the four
        asm volatile("ds_read_b128 v[100:103], %4\n\ts_waitcnt lgkmcnt(0)\n\t"
                     "v_pk_fma_f32 %0, %5, v[100:101], %6 op_sel:[0,1,0] op_sel_hi:[1,1,1]\n\t"
                     "v_pk_fma_f32 %1, %5, v[100:101], %6 op_sel:[0,1,0] op_sel_hi:[1,0,1]\n\t"
                     "v_mov_b32 %2, v100\n\tv_mov_b32 %3, v101"
                     : "=&v"(r2), "=&v"(r3), "=&v"(b0), "=&v"(b1) : "v"(addr), "v"(a), "v"(c) : "memory", "v100", "v101", "v102", "v103");
        float lh, hl, hh;
        asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(lh) : "v"(a[0]), "v"(b1), "v"(c[0]));
        asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(hl) : "v"(a[1]), "v"(b0), "v"(c[1]));
        asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(hh) : "v"(a[1]), "v"(b1), "v"(c[1]));
        wrong[0][0] += r2[0] != lh; wrong[0][1] += r2[1] != hh;
        wrong[1][0] += r3[0] != lh; wrong[1][1] += r3[1] != hl;
        a[0] += 1.0f; c[0] += 0.25f;
    }
#pragma unroll
    for (int f = 0; f < 2; ++f)
#pragma unroll
        for (int h = 0; h < 2; ++h)
            if (wrong[f][h]) atomicAdd(bad + ((f + 2) * 2 + h) * 4 + (lane >> 4), (unsigned long long)wrong[f][h]);
}

template <bool VALU>
__global__ __launch_bounds__(256) void aggressor(const s16x8* __restrict__ src, float* __restrict__ out, int iters) {
    extern __shared__ __attribute__((aligned(16))) s16x8 sm[];                   // 48 KB
    const int tid = threadIdx.x;
    for (int i = tid; i < 12 * 256; i += 256) sm[i] = src[i];
    __syncthreads();
    s16x8 a[6], b[6];
    f32x16 acc[4];
    unsigned u = tid;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 6; ++i) { a[i] = sm[i * 256 + ((tid + it) & 255)]; b[i] = sm[(6 + i) * 256 + ((tid + 7 * it) & 255)]; }
#pragma unroll
        for (int m = 0; m < 24; ++m) {
            acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[m % 6], b[(m / 4) % 6], acc[m & 3], 0, 0, 0);
            if (VALU) {                                 // two ordinary VALU instructions in the shadow of every MFMA
                asm volatile("v_mov_b32 %0, %1" : "=v"(u) : "v"(u));
                asm volatile("v_mov_b32 %0, %1" : "=v"(u) : "v"(u));
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    if (u == 0xdeadbeefu) out[0] = 1.f;
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + tid] = s;
}

int main(int argc, char** argv) {
    const double seconds = argc > 1 ? atof(argv[1]) : 3.0;
    const int agg_wgs = argc > 2 ? atoi(argv[2]) : 512;             // 0: no aggressor
    const bool lds_mode = argc > 3 && atoi(argv[3]) != 0;
    const bool agg_valu = argc > 4 && atoi(argv[4]) != 0;           // 1: the aggressor issues two v_mov_b32 after every MFMA            // 1: coefficient pairs straight from ds_read_b128 (victim_lds)
    std::vector<unsigned short> h(12 * 256 * 8);
    for (auto& v : h) { const unsigned r = (unsigned)rand(); v = (unsigned short)(((r & 1) << 15) | ((126 + ((r >> 1) & 1)) << 7) | ((r >> 2) & 0x7f)); }
    s16x8* d; float* o; unsigned long long* bad;
    CK(hipMalloc(&d, h.size() * 2)); CK(hipMalloc(&o, (size_t)4096 * 256 * 4)); CK(hipMalloc(&bad, 64 * 8)); CK(hipMemset(bad, 0, 64 * 8));
    CK(hipMemcpy(d, h.data(), h.size() * 2, hipMemcpyHostToDevice));
    CK(hipFuncSetAttribute((const void*)aggressor<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 49152));
    CK(hipFuncSetAttribute((const void*)aggressor<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 49152));
    hipStream_t s1, s2; CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    long launches = 0; float ms = 0;
    CK(hipEventRecord(e0, s2));
    while (ms < seconds * 1e3) {
        for (int i = 0; i < 20; ++i) {
            if (agg_wgs && agg_valu) hipLaunchKernelGGL(aggressor<true>, dim3(agg_wgs), dim3(256), 49152, s1, d, o, 300);
            else if (agg_wgs) hipLaunchKernelGGL(aggressor<false>, dim3(agg_wgs), dim3(256), 49152, s1, d, o, 300);
            for (int j = 0; j < 4; ++j) {
                if (lds_mode) hipLaunchKernelGGL(victim_lds, dim3(2048), dim3(256), 0, s2, bad, 400, (float)(launches + j));
                else hipLaunchKernelGGL(victim, dim3(2048), dim3(256), 0, s2, bad, 400, (float)(launches + j));
            }
            launches += 4;
        }
        CK(hipEventRecord(e1, s2)); CK(hipEventSynchronize(e1)); CK(hipStreamSynchronize(s1));
        CK(hipEventElapsedTime(&ms, e0, e1));
    }
    CK(hipDeviceSynchronize());
    unsigned long long hb[64]; CK(hipMemcpy(hb, bad, sizeof hb, hipMemcpyDeviceToHost));
    const char* names[8] = {"plain", "op_sel_hi:[1,0,1] (low half broadcast)", "op_sel:[0,1,0] (high half broadcast)", "op_sel:[0,1,0] op_sel_hi:[1,0,1] (swap)", "addend 0, op_sel_hi:[1,0,0] (low half bcast)", "addend 0, op_sel:[0,1,0] op_sel_hi:[1,1,0]", "pair: op_sel first, its result the addend", "pair: op_sel second, on the dependent one"};
    unsigned long long total = 0;
    printf("%ld %s launches (2048 x 256 threads x 400 rounds) beside %s, %.1f s\n", launches, lds_mode ? "victim_lds" : "victim", !agg_wgs ? "nothing" : agg_valu ? "a bf16 MFMA kernel with v_mov_b32 between its MFMAs, on a second stream" : "a bf16 MFMA kernel (no VALU in its loop) on a second stream", ms * 1e-3);
    for (int f = 0; f < 8; ++f)
        for (int hlf = 0; hlf < 2; ++hlf) {
            const unsigned long long* q = hb + (f * 2 + hlf) * 4;
            printf("  %-44s %s result: wrong in lanes 0-15 %llu, 16-31 %llu, 32-47 %llu, 48-63 %llu\n", names[f], hlf ? "high" : "low ", q[0], q[1], q[2], q[3]);
            total += q[0] + q[1] + q[2] + q[3];
        }
    printf("total wrong results: %llu\n", total);
    return 0;
}
