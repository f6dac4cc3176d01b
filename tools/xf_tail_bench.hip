// Micro-benchmark of the fused transformer tail (csrc/xf_tail.hip) on the demo step's shapes; experiment switches:
// -DXT_EXP_NOGELU (identity instead of GELU), -DXT_EXP_NOSYNC (no slab barrier / DMA: compute on stale LDS), XT_WG_PER_CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../mocha_sigasia2023_amd/csrc/xf_tail.hip"
using namespace mocha;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)
int main(int argc, char** argv) {
    CK(xf_tail_init());
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int cfg = 0; cfg < 2; ++cfg) {
        const int M = cfg ? 52650 : 105300, Kin = cfg ? 1024 : 512;
        auto dev = [&](size_t n, float sc) { std::vector<float> h(n); for (auto& v : h) v = sc * ((rand() & 0xffff) / 32768.f - 1.f); float* d; CK(hipMalloc(&d, n * 4)); CK(hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice)); return d; };
        XfTailParams p{dev((size_t)M * Kin, 1.f), Kin, dev(256 * Kin, 0.05f), dev(256, 0.1f), dev((size_t)M * 256, 1.f), dev(512 * 256, 0.05f), dev(512, 0.1f),
                       dev(256 * 512, 0.05f), dev(256, 0.1f), dev((size_t)M * 256, 0.f), M};
        for (int i = 0; i < 3; ++i) CK(launch_xf_tail(p, 0));
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0, 0));
        const int it = 20;
        for (int i = 0; i < it; ++i) CK(launch_xf_tail(p, 0));
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= it;
        const double fl = 2.0 * M * (256.0 * Kin + 2.0 * 256 * 512);
        printf("M=%6d Kin=%4d: %.1f us  %.1f TFLOP/s\n", M, Kin, ms * 1e3, fl / ms / 1e9);
    }
    return 0;
}
