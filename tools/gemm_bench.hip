// Micro-benchmark + self-check of the fp32 MFMA GEMM on the shapes of the MOCHA path.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I mocha_sigasia2023_amd/csrc tools/gemm_bench.hip \
//        mocha_sigasia2023_amd/csrc/gemm_f32.o -o gpurun_out/gemm_bench      (tools/build_gemm_bench.sh)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include <algorithm>
#include <string>
#include <cstring>
#include "kernels.h"
using namespace mocha;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

__global__ void ref_gemm(const float* A, const float* W, double* C, int M, int N, int K) {
    int n = blockIdx.x * 16 + threadIdx.x, m = blockIdx.y * 16 + threadIdx.y;
    if (m >= M || n >= N) return;
    double a = 0;
    for (int k = 0; k < K; ++k) a += (double)A[(size_t)m * K + k] * W[(size_t)n * K + k];
    C[(size_t)m * N + n] = a;
}

struct Shape { const char* name; int M, N, K; int gather; int T_out, V, ntaps, pad, stride, R, T_full, tshift, Cc, T_src; int lda; };

int main(int argc, char** argv) {
    int B = argc > 1 ? atoi(argv[1]) : 585;
    int iters = argc > 2 ? atoi(argv[2]) : 20;
    int check = argc > 3 ? atoi(argv[3]) : 1;
    int mode = argc > 4 ? atoi(argv[4]) : 0;      // 0 = exact f32 MFMA (gemm_f32.hip), 36 = bf16 x 3 planes (gemm_x3.hip), 22 = fp16 x 2 planes (gemm_h2.hip)
    CK(gemm_init()); CK(gemm_x3_init()); CK(gemm_h2_init()); CK(gemm_x3r_init());
    std::vector<Shape> shapes = {
        {"enc.qkv      ", B * 90, 1536, 256, 0},
        {"xf.out512    ", B * 90, 256, 512, 0},
        {"xf.ff1       ", B * 90, 512, 256, 0},
        {"xf.ff1 1024  ", B * 90, 1024, 256, 0},
        {"dec.q        ", B * 90, 1024, 256, 0},
        {"dec.out1024  ", B * 90, 256, 1024, 0},
        {"emb.gcn_joint", B * 360, 256, 192, 0},
        {"emb.tcn_pool ", B * 90, 256, 1280, 1, 15, 6, 5, 2, 4, 4, 60, 0, 256, 60, 256},
        {"emb.tcn_body ", B * 90, 256, 768, 1, 15, 6, 3, 1, 1, 1, 15, 0, 256, 15, 256},
        {"mot.tcn_joint", B * 1440, 64, 320, 1, 60, 24, 5, 2, 1, 1, 60, 2, 64, 15, 64},
        {"mot.gcn_joint", B * 90, 192, 256, 0},
        {"cvae M=182   ", 182, 768, 256, 0},
        {"cvae M=364   ", 364, 768, 256, 0},
        {"cvae M=90    ", 90, 768, 256, 0},
        {"match 585    ", 585, 585, 23040, 0},
        {"square 4096  ", 4096, 4096, 4096, 0},
    };
    if (getenv("MOCHA_BENCH_SHAPES")) {            // "M,N,K;M,N,K;..." plain shapes
        shapes.clear();
        static std::vector<std::string> names;
        const char* q = getenv("MOCHA_BENCH_SHAPES");
        while (*q) {
            int m, n, k, used = 0;
            if (sscanf(q, "%d,%d,%d%n", &m, &n, &k, &used) != 3) break;
            shapes.push_back({"custom       ", m, n, k, 0});
            q += used; if (*q == ';') ++q;
        }
    }
    if (getenv("MOCHA_BENCH_KSWEEP")) {            // time vs K at fixed M, N: intercept = per-launch fixed cost
        shapes.clear();
        static const int ks[] = {32, 64, 128, 256, 384, 512, 768, 1024, 1536, 2048};
        const int n = atoi(getenv("MOCHA_BENCH_KSWEEP"));
        for (int k : ks) shapes.push_back({"ksweep       ", B * 90, n, k, 0});
    }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (auto& sh : shapes) {
        size_t a_rows = sh.gather ? (size_t)(sh.M / (sh.T_out * sh.V)) * sh.T_src * sh.V : sh.M;
        int lda = sh.gather ? sh.lda : sh.K;
        size_t na = a_rows * lda, nw = (size_t)sh.N * sh.K, nc = (size_t)sh.M * sh.N;
        std::vector<float> ha(na), hw(nw);
        const bool zero = getenv("MOCHA_BENCH_ZERO") != nullptr;      // zero operands: the clock the chip holds without data toggling
        for (auto& v : ha) v = zero ? 0.f : (float)rand() / RAND_MAX * 2 - 1;
        for (auto& v : hw) v = zero ? 0.f : (float)rand() / RAND_MAX * 2 - 1;
        float *dA, *dW, *dC; double* dR;
        CK(hipMalloc(&dA, na * 4)); CK(hipMalloc(&dW, nw * 4)); CK(hipMalloc(&dC, nc * 4 * (sh.M == 585 && sh.N == 585 ? 16 : 1)));
        CK(hipMemcpy(dA, ha.data(), na * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dW, hw.data(), nw * 4, hipMemcpyHostToDevice));
        unsigned short* dWs = nullptr;
        if (mode == 36 || mode == 37) { CK(hipMalloc(&dWs, gemm_x3_packed_elems(sh.N, sh.K) * 2)); CK(launch_pack_x3(dW, sh.N, sh.K, dWs, 0)); }
        GemmParams p; p.Wsplit = dWs; p.A = dA; p.W = dW; p.C = dC; p.M = sh.M; p.N = sh.N; p.K = sh.K; p.lda = lda; p.ldc = sh.N;
        if (sh.gather) { p.gather = 1; p.T_out = sh.T_out; p.V = sh.V; p.ntaps = sh.ntaps; p.pad = sh.pad; p.stride = sh.stride; p.R = sh.R;
                         p.T_full = sh.T_full; p.tshift = sh.tshift; p.Cc = sh.Cc; p.T_src = sh.T_src; p.ascale = sh.R > 1 ? 0.25f : 1.f; }
        if (sh.M == 585 && sh.N == 585) { p.ksplit = 16; p.slab_stride = (long long)sh.M * sh.N; }
        if (getenv("MOCHA_BENCH_TILE64")) p.tile64_below = atoi(getenv("MOCHA_BENCH_TILE64"));      // 64 x 64 tiles for mid-size launches (gemm_x3.hip: x3_tile64)
        if (getenv("MOCHA_BENCH_PERSISTENT")) p.persistent = atoi(getenv("MOCHA_BENCH_PERSISTENT"));      // 0: every launch on the one-shot grid (mocha_gemm_x3)
        if (getenv("MOCHA_BENCH_ALRELU")) p.a_lrelu = 1;               // LeakyReLU on the A operand as it is split: what an A-operand prologue costs the K loop
        const bool x3 = (mode == 36 || mode == 37) && gemm_x3_supports(p);
        const bool x3r_bias = getenv("MOCHA_BENCH_BIASGELU") != nullptr;       // bias + GELU epilogue (ff1)
        float* dbias = nullptr;
        if (x3r_bias) { std::vector<float> hb(sh.N); for (auto& v : hb) v = (float)rand() / RAND_MAX - 0.5f; CK(hipMalloc(&dbias, sh.N * 4)); CK(hipMemcpy(dbias, hb.data(), sh.N * 4, hipMemcpyHostToDevice)); p.bias = dbias; p.act = 1; }
        const bool x3r = mode == 37 && x3 && gemm_x3r_supports(p);
        const int x3r_grid = getenv("MOCHA_BENCH_X3R_GRID") ? atoi(getenv("MOCHA_BENCH_X3R_GRID")) : 0;
        p.rows_per_win = sh.M % 90 == 0 ? 90 : 1024;
        const bool h2 = mode == 22 && !sh.gather && gemm_h2_supports(p);
        float* daux = nullptr;                                         // [N] inverse weight scales | bias [N] | per-window activation bounds | per-window output bounds
        float* dres = nullptr;
        const bool epi = getenv("MOCHA_BENCH_EPI") != nullptr;        // bias + residual epilogue (out_proj / ff2); the float64 check is skipped
        const int rpw = sh.M % 90 == 0 ? 90 : 1024;                    // rows per "window" of the two-plane fp16 engine's scales
        const long long nwin = (sh.M + rpw - 1) / rpw;
        CK(hipMalloc(&daux, ((size_t)2 * sh.N + 2 * nwin) * 4)); CK(hipMemset(daux, 0, ((size_t)2 * sh.N + 2 * nwin) * 4));
        if (epi) { CK(hipMalloc(&dres, nc * 4)); CK(hipMemset(dres, 0, nc * 4)); p.bias = daux + sh.N; p.residual = dres; p.ldr = sh.N; }
        if (h2) {
            unsigned short* dWh = nullptr;
            CK(hipMalloc(&dWh, gemm_h2_packed_elems(sh.N, sh.K) * 2)); CK(launch_pack_h2(dW, sh.N, sh.K, dWh, daux, 0));
            for (long long w0 = 0; w0 < nwin; w0 += 65535) {
                const long long nw = std::min<long long>(65535, nwin - w0), full = (w0 + nw == nwin && sh.M % rpw) ? nw - 1 : nw;
                if (full > 0) CK(launch_absmax(dA + (size_t)w0 * rpw * lda, full, (long long)rpw * lda, daux + 2 * sh.N + w0, 0));
                if (full < nw) CK(launch_absmax(dA + (size_t)(w0 + full) * rpw * lda, 1, (long long)(sh.M % rpw) * lda, daux + 2 * sh.N + w0 + full, 0));
            }
            p.Wh2 = dWh; p.w_inv = daux; p.a_amax = daux + 2 * sh.N; p.rows_per_win = rpw;
            if (getenv("MOCHA_BENCH_CAMAX")) p.c_amax = daux + 2 * sh.N + nwin;      // the epilogue's per-window maxima of what it stores
        }
        const bool rezero = h2 && p.c_amax && getenv("MOCHA_BENCH_REZERO");      // the output bound starts from zero at every launch, as in the pipeline
        auto run = [&]() { if (rezero) (void)hipMemsetAsync(p.c_amax, 0, nwin * 4, 0); return h2 ? launch_gemm_h2(p, 0) : x3r ? launch_gemm_x3r(p, 0, x3r_grid) : x3 ? launch_gemm_x3(p, 0) : launch_gemm(p, 0); };
        long long* dstamp = nullptr;
        if (getenv("MOCHA_BENCH_STAMPS") && x3) { CK(hipMalloc(&dstamp, (size_t)65536 * 32)); CK(hipMemset(dstamp, 0, (size_t)65536 * 32)); p.wsub = (const float*)dstamp; }
        for (int i = 0; i < 3; ++i) CK(run());
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0, 0));
        for (int i = 0; i < iters; ++i) CK(run());
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= iters;
        double tf = 2.0 * sh.M * sh.N * sh.K / (ms * 1e-3) / 1e12;
        double err = -1, rms = -1;
        long long bitdiff = -1;
        if (x3r && check) {                                              // bit for bit against mocha_gemm_x3 on the same operands
            std::vector<float> h1((size_t)sh.M * sh.N), h2v((size_t)sh.M * sh.N);
            CK(hipMemcpy(h1.data(), dC, nc * 4, hipMemcpyDeviceToHost));
            CK(hipMemset(dC, 0xff, nc * 4));
            CK(launch_gemm_x3(p, 0)); CK(hipDeviceSynchronize());
            CK(hipMemcpy(h2v.data(), dC, nc * 4, hipMemcpyDeviceToHost));
            bitdiff = 0;
            for (size_t i = 0; i < nc; ++i) bitdiff += memcmp(&h1[i], &h2v[i], 4) != 0;
            CK(hipMemset(dC, 0xff, nc * 4)); CK(run()); CK(hipDeviceSynchronize());
        }
        if (check && !epi && !x3r_bias && !sh.gather && p.ksplit == 1 && (double)sh.M * sh.N * sh.K < 3e11) {
            CK(hipMalloc(&dR, nc * 8));
            hipLaunchKernelGGL(ref_gemm, dim3((sh.N + 15) / 16, (sh.M + 15) / 16), dim3(16, 16), 0, 0, dA, dW, dR, sh.M, sh.N, sh.K);
            std::vector<float> hc(nc); std::vector<double> hr(nc);
            CK(hipMemcpy(hc.data(), dC, nc * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(hr.data(), dR, nc * 8, hipMemcpyDeviceToHost));
            err = 0; double se = 0;
            for (size_t i = 0; i < nc; ++i) { const double d = (double)hc[i] - hr[i]; err = fmax(err, fabs(d)); se += d * d; }
            rms = sqrt(se / nc);
            if (getenv("MOCHA_BENCH_DIAG")) {          // where are the wrong elements?
                long bad = 0; std::vector<long> byrow(128, 0), bycol(128, 0); long first = -1;
                std::vector<long> bymt((sh.M + 127) / 128, 0);
                for (size_t i = 0; i < nc; ++i) {
                    const double d = fabs((double)hc[i] - hr[i]);
                    if (!(d < 1e-2)) { ++bad; if (first < 0) first = (long)i; byrow[(i / sh.N) % 128]++; bycol[(i % sh.N) % 128]++; bymt[(i / sh.N) / 128]++; }
                }
                long tiles_bad = 0; for (long v : bymt) tiles_bad += v > 0;
                printf("  bad=%ld first=(%ld,%ld) bad m-tiles=%ld of %zu\n  rows-in-tile:", bad, first / sh.N, first % sh.N, tiles_bad, bymt.size());
                for (int r = 0; r < 128; ++r) if (byrow[r]) printf(" %d:%ld", r, byrow[r]);
                printf("\n  cols-in-tile:");
                for (int r = 0; r < 128; ++r) if (bycol[r]) printf(" %d:%ld", r, bycol[r]);
                printf("\n  bad m-tiles:"); int shown = 0;
                for (size_t t = 0; t < bymt.size() && shown < 40; ++t) if (bymt[t]) { printf(" %zu:%ld", t, bymt[t]); ++shown; }
                printf("\n");
            }
            CK(hipFree(dR));
        }
        if (dstamp && x3r) {             // gemm_x3r.hip built with -DX3R_STAMPS: per wave {issue, wait, barrier, -, unit changes, -, total cycles, realtime ticks}
            std::vector<long long> h((size_t)256 * 4 * 8);
            CK(hipMemcpy(h.data(), dstamp, h.size() * 8, hipMemcpyDeviceToHost));
            double a[8] = {0, 0, 0, 0, 0, 0, 0, 0}; long n = 0;
            for (int w = 0; w < 256 * 4; ++w) { if (!h[8 * (size_t)w + 6]) continue; for (int i = 0; i < 8; ++i) a[i] += h[8 * (size_t)w + i]; ++n; }
            if (n) printf("  %ld waves: issue (MFMAs, reads, copies, store) %.0f  counted wait %.0f  barrier %.0f  unit changes %.0f  other %.0f  of %.0f shader cycles per wave; %.1f us = %.2f GHz\n",
                          n, a[0] / n, a[1] / n, a[2] / n, a[4] / n, (a[3] + a[5]) / n, a[6] / n, a[7] / n / 100.0, a[6] / a[7] / 10.0);
            CK(hipFree(dstamp)); dstamp = nullptr;
        }
        if (dstamp) {
            const int wgs = ((sh.M + 127) / 128 + 7) / 8 * 8 * ((sh.N + 127) / 128);
            std::vector<long long> h((size_t)wgs * 6);
            CK(hipMemcpy(h.data(), dstamp, h.size() * 8, hipMemcpyDeviceToHost));
            double pro = 0, loop = 0, epi = 0, life_rt = 0; long n = 0; long long tmin = 0, tmax = 0;
            for (int w = 0; w < wgs; ++w) {
                const long long* d = &h[6 * (size_t)w];
                if (!d[0]) continue;
                pro += d[1] - d[0]; loop += d[2] - d[1]; epi += d[3] - d[2]; life_rt += d[5] - d[4]; ++n;
                if (!tmin || d[4] < tmin) tmin = d[4];
                if (d[5] > tmax) tmax = d[5];
            }
            const double span_us = (tmax - tmin) / 100.0, life_us = life_rt / n / 100.0;
            printf("  %ld workgroups: prologue %.0f  K loop %.0f (%.1f per step)  epilogue %.0f shader cycles; lifetime %.1f us = %.2f GHz; first start to last end %.1f us; slot occupancy %.2f\n",
                   n, pro / n, loop / n, loop / n / (sh.K / 16), epi / n, life_us, (pro + loop + epi) / n / life_us / 1e3, span_us, n * life_us / 768.0 / span_us);
            CK(hipFree(dstamp));
        }
        printf("%s M=%7d N=%5d K=%5d  %9.1f us  %7.2f TFLOP/s  (%.1f%% of 157.3)  %s maxerr=%.3g rms=%.3g\n", sh.name, sh.M, sh.N, sh.K, ms * 1e3, tf, tf / 157.3 * 100, h2 ? "h2 " : x3r ? "x3r" : x3 ? "x3 " : "f32", err, rms);
        if (bitdiff >= 0) printf("    elements that differ from mocha_gemm_x3 in any bit: %lld of %zu\n", bitdiff, nc);
        if (dbias) CK(hipFree(dbias));
        CK(hipFree(dA)); CK(hipFree(dW)); CK(hipFree(dC)); if (dWs) CK(hipFree(dWs)); CK(hipFree(daux)); if (dres) CK(hipFree(dres)); if (p.Wh2) CK(hipFree((void*)p.Wh2));
    }
    return 0;
}
