// Self-check of the attention kernel variants against a host reference (diagnostic tool).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include "kernels.h"
using namespace mocha;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)
int main() {
    struct Case { int dh, nq, nk, heads, B; } cases[] = {{128, 90, 90, 4, 3}, {256, 90, 90, 4, 2}, {64, 90, 90, 4, 2}, {64, 90, 181, 4, 2}, {64, 182, 182, 4, 2}, {64, 100, 100, 4, 1}};
    for (auto cs : cases) {
        const int inner = cs.heads * cs.dh;
        size_t nqe = (size_t)cs.B * cs.nq * inner, nke = (size_t)cs.B * cs.nk * inner;
        std::vector<float> q(nqe), k(nke), v(nke), o(nqe), ref(nqe);
        for (auto& x : q) x = (float)(rand() % 2001 - 1000) / 500.f;
        for (auto& x : k) x = (float)(rand() % 2001 - 1000) / 500.f;
        for (auto& x : v) x = (float)(rand() % 2001 - 1000) / 500.f;
        const float scale = 1.f / sqrtf((float)cs.dh);
        for (int b = 0; b < cs.B; ++b) for (int h = 0; h < cs.heads; ++h) for (int i = 0; i < cs.nq; ++i) {
            std::vector<double> s(cs.nk); double mx = -1e30;
            for (int j = 0; j < cs.nk; ++j) { double a = 0; for (int d = 0; d < cs.dh; ++d) a += (double)q[((size_t)b * cs.nq + i) * inner + h * cs.dh + d] * k[((size_t)b * cs.nk + j) * inner + h * cs.dh + d]; s[j] = a * scale; mx = fmax(mx, s[j]); }
            double sum = 0; for (auto& x : s) { x = exp(x - mx); sum += x; }
            for (int d = 0; d < cs.dh; ++d) { double a = 0; for (int j = 0; j < cs.nk; ++j) a += s[j] * v[((size_t)b * cs.nk + j) * inner + h * cs.dh + d]; ref[((size_t)b * cs.nq + i) * inner + h * cs.dh + d] = (float)(a / sum); }
        }
        float *dq, *dk, *dv, *dout;
        CK(hipMalloc(&dq, nqe * 4)); CK(hipMalloc(&dk, nke * 4)); CK(hipMalloc(&dv, nke * 4)); CK(hipMalloc(&dout, nqe * 4));
        CK(hipMemcpy(dq, q.data(), nqe * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dk, k.data(), nke * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dv, v.data(), nke * 4, hipMemcpyHostToDevice));
        CK(hipMemset(dout, 0, nqe * 4));
        AttnParams p{dq, dk, dv, dout, inner, inner, inner, inner, cs.B, cs.heads, cs.dh, cs.nq, cs.nk, scale};
        CK(launch_attention(p, 0)); CK(hipDeviceSynchronize());
        CK(hipMemcpy(o.data(), dout, nqe * 4, hipMemcpyDeviceToHost));
        double err = 0; int worst = -1;
        for (size_t i = 0; i < nqe; ++i) { double e = fabs((double)o[i] - ref[i]); if (e > err) { err = e; worst = (int)i; } }
        printf("dh=%3d nq=%3d nk=%3d: max err %.3e (worst at row %d col %d)\n", cs.dh, cs.nq, cs.nk, err, worst / inner % cs.nq, worst % inner);
        if (cs.nq == cs.nk) {      // interleaved [q | k | v] rows of 3*inner floats, as the in_proj GEMM writes them
            std::vector<float> qkv((size_t)cs.B * cs.nq * 3 * inner);
            for (size_t r = 0; r < (size_t)cs.B * cs.nq; ++r) for (int cidx = 0; cidx < inner; ++cidx) {
                qkv[r * 3 * inner + cidx] = q[r * inner + cidx]; qkv[r * 3 * inner + inner + cidx] = k[r * inner + cidx]; qkv[r * 3 * inner + 2 * inner + cidx] = v[r * inner + cidx]; }
            float* dqkv; CK(hipMalloc(&dqkv, qkv.size() * 4)); CK(hipMemcpy(dqkv, qkv.data(), qkv.size() * 4, hipMemcpyHostToDevice));
            CK(hipMemset(dout, 0, nqe * 4));
            AttnParams p2{dqkv, dqkv + inner, dqkv + 2 * inner, dout, 3 * inner, 3 * inner, 3 * inner, inner, cs.B, cs.heads, cs.dh, cs.nq, cs.nk, scale};
            CK(launch_attention(p2, 0)); CK(hipDeviceSynchronize());
            CK(hipMemcpy(o.data(), dout, nqe * 4, hipMemcpyDeviceToHost));
            double e2 = 0; for (size_t i = 0; i < nqe; ++i) e2 = fmax(e2, fabs((double)o[i] - ref[i]));
            printf("   interleaved qkv layout: max err %.3e\n", e2);
            CK(hipFree(dqkv));
        }
        CK(hipFree(dq)); CK(hipFree(dk)); CK(hipFree(dv)); CK(hipFree(dout));
    }
    return 0;
}
