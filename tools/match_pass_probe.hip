// Floors and candidates of the many-query bf16 coarse pass:  Q queries x N bank rows x 23 040, bf16.
//   old        mocha_match_gemm_bf16_dma<4,1>   (match_mfma.hip: 128 x 128 tiles, both operands through the LDS-DMA ring, K split 8)
//   p256/<v>   mocha_match_pass256              (match_pass.hip: 128 x 256 tiles, bank straight into registers, K split 16); v = variant bits:
//              prefetch depth 3..6 (+16: non-temporal bank loads, +256: FILL ONLY - the same loads, barriers and ring, no LDS reads, no MFMA)
// Every configuration is timed COLD-CLEAN (a 1 GiB READ runs between launches: the caches hold clean foreign lines, nothing of the bank - the
// state the pass finds inside mocha_characterize), COLD-DIRTY (a 1 GiB memset instead: the caches are full of lines that must be written
// back while the pass reads - an upper bound no pipeline state reaches) and WARM (back to back: a 189 MB bank stays resident in the
// 256 MB Infinity Cache).  "stream" rows: the memory system's own ceiling on the same buffer (1 KB contiguous per wave-level load).
// HIP events around single launches; the algorithmic bytes are bank + queries once.  Sum over the K slices of S is compared
// between the kernels.
//   build:  tools/build_match_probe.sh      run:  tools/bin/match_pass_probe [Q] [N]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include <algorithm>
#include "kernels.h"

using namespace mocha;
#define CK(x) do { hipError_t e__ = (x); if (e__ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e__)); exit(1); } } while (0)

__global__ void fill_bf16(unsigned short* p, size_t n, unsigned seed) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned h = (unsigned)i * 2654435761u ^ seed; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        const float v = ((h & 0xffff) / 65536.0f - 0.5f) * 2.0f;                    // uniform (-1, 1)
        p[i] = (unsigned short)(__float_as_uint(v) >> 16);
    }
}
// the memory system's own ceiling on this buffer: every wave-level load 1 KB contiguous, 8 loads in flight per lane (the few-query scan's pattern)
typedef unsigned int pu32x4 __attribute__((ext_vector_type(4)));
template <bool NT>
__global__ __launch_bounds__(256) void stream_read(const pu32x4* __restrict__ p, size_t n16, unsigned* __restrict__ sink) {
    unsigned a = 0;
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + 7 * stride < n16; i += 8 * stride) {
        pu32x4 v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = NT ? __builtin_nontemporal_load(p + i + k * stride) : p[i + k * stride];
#pragma unroll
        for (int k = 0; k < 8; ++k) a ^= v[k].x ^ v[k].y ^ v[k].z ^ v[k].w;
    }
    for (; i < n16; i += stride) { const pu32x4 v = p[i]; a ^= v.x ^ v.y ^ v.z ^ v.w; }
    if (a == 0x12345678u) sink[0] = a;
}
__global__ void slab_sum(const float* S, int ksplit, size_t slab, float* out, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float a = 0.f;
    for (int z = 0; z < ksplit; ++z) a += S[(size_t)z * slab + i];
    out[i] = a;
}

int main(int argc, char** argv) {
    const int Q = argc > 1 ? atoi(argv[1]) : 128;
    const long long N = argc > 2 ? atoll(argv[2]) : 4096;
    const int D = 23040;
    CK(match_mfma_init());
    unsigned short *A, *B; float *S, *ref, *got; char* flush;
    const size_t flush_bytes = (size_t)1 << 30;
    CK(hipMalloc(&A, (size_t)2 * Q * D * 2 + 4096)); CK(hipMalloc(&B, (size_t)N * D * 2));
    CK(hipMalloc(&S, (size_t)16 * Q * N * 4)); CK(hipMalloc(&ref, (size_t)Q * N * 4)); CK(hipMalloc(&got, (size_t)Q * N * 4));
    CK(hipMalloc(&flush, flush_bytes));
    fill_bf16<<<2048, 256>>>(A, (size_t)2 * Q * D, 1u); fill_bf16<<<4096, 256>>>(B, (size_t)N * D, 2u);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const double bytes = 2.0 * ((double)N + Q) * D;

    unsigned short* Bt; unsigned* sink;
    CK(hipMalloc(&Bt, match_tile32_elems(N, D) * 2)); CK(hipMalloc(&sink, 64));
    CK(launch_tile32_bf16(B, Bt, N, D, 0)); CK(hipDeviceSynchronize());
    struct Cfg { const char* name; int variant; int planes = 1; int tiled = 0; };          // variant < 0: the round-4 kernel; -2 / -3: pure stream (plain / nt)
    const Cfg cfgs[] = {{"stream 1 KB/instr", -2}, {"stream 1 KB/instr nt", -3}, {"old dma<4,1> ksplit 8", -1}, {"old dma<4,1> nt bank", -4},
                        {"p256 pfs5 tiled", 5, 1, 1}, {"p256 pfs6 tiled", 6, 1, 1}, {"p256 pfs5 tiled nt", 5 + 16, 1, 1}, {"p256 pfs5 tiled FILL", 5 + 256, 1, 1},
                        {"p256 pfs5 tiled nt FILL", 5 + 16 + 256, 1, 1}, {"p256 2pl pfs3 tiled", 3, 2, 1},
                        {"p256 pfs3", 3}, {"p256 pfs4", 4}, {"p256 pfs5", 5}, {"p256 pfs6", 6},
                        {"p256 pfs4 nt", 4 + 16}, {"p256 pfs5 nt", 5 + 16}, {"p256 pfs6 nt", 6 + 16},
                        {"old dma<3,2> 2 planes", -1, 2}, {"p256 2pl pfs2", 2, 2}, {"p256 2pl pfs3", 3, 2}, {"p256 2pl pfs4", 4, 2}, {"p256 2pl pfs3 nt", 3 + 16, 2},
                        {"p256 2pl pfs4 nt", 4 + 16, 2},
                        {"p256 pfs5 FILL", 5 + 256}, {"p256 pfs6 FILL", 6 + 256}, {"p256 pfs5 nt FILL", 5 + 16 + 256}, {"p256 pfs6 nt FILL", 6 + 16 + 256}};
    const int k_old = match_bf16_ksplit(Q, N), k_new = match_pass256_ksplit(Q, N);
    printf("Q = %d, N = %lld, D = %d: algorithmic %.1f MB; K split old %d, new %d\n", Q, N, D, bytes / 1e6, k_old, k_new);
    bool have_ref = false; int ref_planes = 1;
    for (const Cfg& c : cfgs) {
        auto launch = [&]() {
            if (c.variant == -2) stream_read<false><<<2048, 256>>>((const pu32x4*)B, (size_t)N * D / 8, sink);
            else if (c.variant == -3) stream_read<true><<<2048, 256>>>((const pu32x4*)B, (size_t)N * D / 8, sink);
            else if (c.variant < 0) CK(launch_match_gemm_bf16(A, B, S, Q, N, D, k_old, 0, c.planes, nullptr, c.variant == -4 ? 1 : 0));
            else CK(launch_match_pass256(A, B, S, Q, N, D, k_new, 0, c.variant, c.planes, c.tiled ? Bt : nullptr));
        };
        launch(); CK(hipDeviceSynchronize());
        // correctness against the first configuration (not for fill-only runs)
        double err = -1.0;
        if (c.planes != ref_planes) { have_ref = false; ref_planes = c.planes; }
        if (c.variant >= 0 ? !(c.variant & 256) : (c.variant == -1 || c.variant == -4)) {
            const int ks = c.variant < 0 ? k_old : k_new;
            slab_sum<<<(unsigned)(((size_t)Q * N + 255) / 256), 256>>>(S, ks, (size_t)Q * N, have_ref ? got : ref, (size_t)Q * N);
            CK(hipDeviceSynchronize());
            if (have_ref) {
                std::vector<float> a((size_t)Q * N), b((size_t)Q * N);
                CK(hipMemcpy(a.data(), ref, a.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(b.data(), got, b.size() * 4, hipMemcpyDeviceToHost));
                err = 0.0; double mx = 0.0;
                for (size_t i = 0; i < a.size(); ++i) { err = std::max(err, (double)fabsf(a[i] - b[i])); mx = std::max(mx, (double)fabsf(a[i])); }
                err /= mx;
            }
            have_ref = true;
        }
        std::vector<float> cold, warm, clean;
        for (int r = 0; r < 7; ++r) {
            CK(hipMemsetAsync(flush, r, flush_bytes, 0));
            CK(hipEventRecord(e0, 0)); launch(); CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); cold.push_back(ms * 1e3f);
        }
        for (int r = 0; r < 7; ++r) {                            // caches full of CLEAN foreign lines: a 1 GiB read, nothing to write back
            stream_read<false><<<2048, 256>>>((const pu32x4*)flush, flush_bytes / 16, sink);
            CK(hipEventRecord(e0, 0)); launch(); CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); clean.push_back(ms * 1e3f);
        }
        std::sort(clean.begin(), clean.end());
        for (int r = 0; r < 12; ++r) {
            CK(hipEventRecord(e0, 0)); launch(); CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (r >= 2) warm.push_back(ms * 1e3f);
        }
        std::sort(cold.begin(), cold.end()); std::sort(warm.begin(), warm.end());
        const float cm = cold[cold.size() / 2], wm = warm[warm.size() / 2];
        const float km = clean[clean.size() / 2];
        printf("%-24s cold-clean %6.1f us (%.2f TB/s, %.2f of 8)   cold-dirty %6.1f us (%.2f TB/s)   warm %6.1f us (%.2f TB/s)   rel err vs first %s%.2e\n", c.name,
               km, bytes / km / 1e6, bytes / km / 8e6, cm, bytes / cm / 1e6, wm, bytes / wm / 1e6, err < 0 ? "n/a " : "", err < 0 ? 0.0 : err);
        fflush(stdout);
    }
    return 0;
}
