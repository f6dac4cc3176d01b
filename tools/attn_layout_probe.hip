// Does the qkv layout limit the encoder's attention?  mocha_attention_x3<128> on the pipeline's layout (rows of 3072 floats: q | k | v
// of all 8 heads interleaved per token, a head's 512-byte segments 12 KB apart) against a head-major layout (every (window, head)
// owns 90 contiguous rows of q | k | v = 1.5 KB each), same bytes, same arithmetic.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I mocha_sigasia2023_amd/csrc -c tools/attn_layout_probe.hip -o tools/bin/attn_layout_probe.o
//        && hipcc --offload-arch=gfx950 tools/bin/attn_layout_probe.o mocha_sigasia2023_amd/csrc/attention_x3.o -o tools/bin/attn_layout_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "kernels.h"
using namespace mocha;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

int main() {
    const int B = 585, H = 8, DH = 128, T = 90;
    const size_t n = (size_t)B * T * 3 * H * DH;
    std::vector<float> h(n);
    for (auto& v : h) v = (float)rand() / RAND_MAX - 0.5f;
    float *qkv, *out;
    CK(hipMalloc(&qkv, n * 4)); CK(hipMalloc(&out, (size_t)B * T * H * DH * 4));
    CK(hipMemcpy(qkv, h.data(), n * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int layout = 0; layout < 2; ++layout) {
        AttnParams a{};
        if (layout == 0) { a.q = qkv; a.k = qkv + H * DH; a.v = qkv + 2 * H * DH; a.ldq = a.ldk = a.ldv = 3 * H * DH; a.B = B; a.heads = H; a.ldo = H * DH; }
        else { a.q = qkv; a.k = qkv + DH; a.v = qkv + 2 * DH; a.ldq = a.ldk = a.ldv = 3 * DH; a.B = B * H; a.heads = 1; a.ldo = DH; }
        a.out = out; a.dh = DH; a.nq = T; a.nk = T; a.scale = 0.088f;
        for (int i = 0; i < 3; ++i) CK(launch_attention_x3(a, 0));
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0, 0));
        for (int i = 0; i < 20; ++i) CK(launch_attention_x3(a, 0));
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double bytes = (double)n * 4 + (double)B * T * H * DH * 4;
        printf("%-52s %7.1f us per launch   %6.0f GB/s\n", layout == 0 ? "token-major qkv (rows of 3072 floats, as shipped)" : "head-major qkv (90 x 384 floats per (window, head))",
               ms / 20 * 1e3, bytes / (ms / 20 * 1e-3) / 1e9);
    }
    return 0;
}
