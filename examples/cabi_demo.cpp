// A host program on the C ABI alone - no Python, no torch: what a C / C++ / cgo / JNI caller of libmocha_hip.so does.
//   cabi_demo <weights.bin> <inputs.bin> <out.bin>
// weights.bin (written by examples/export_for_cabi_demo.py from a state_dict with the reference's key names):
//   "MOCHAW01", int32 layout (0 'mocha' 24 joints / 1 'mixamo' 22), int32 count, then per entry
//   int32 name_len, name bytes, int32 ndim, int64 shape[ndim], float32 data
// inputs.bin: int32 B, int32 V, float32 src_X[B,60,V,15], cha_X[B,60,V,15], cnt_mean[90,256], cnt_std[90,256]
// out.bin: float32 Y_forward[B,60,V,15]   Generator.forward(src, cha)                       (model.py:82-106)
//          float32 Y_char[B,60,V,15], int32 idx[B]   the demo sequence: encode(cha) -> bank -> characterize(src)
//                                                    (test_fullframework.py:190-194, 293-298, 440-443, 465-467)
// tests/test_native_host.py builds it, runs it and compares out.bin bit for bit with the Python host's results.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "mocha_hip.h"

#define HIP_OK(x) do { hipError_t e__ = (x); if (e__ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e__)); return 2; } } while (0)
#define MOCHA_OK_(ctx, x) do { int rc__ = (x); if (rc__) { fprintf(stderr, "%s failed (%d): %s\n", #x, rc__, mocha_last_error(ctx)); return 3; } } while (0)

template <class T> static bool rd(FILE* f, T* p, size_t n) { return fread(p, sizeof(T), n, f) == n; }

int main(int argc, char** argv) {
    if (argc != 4) { fprintf(stderr, "usage: %s weights.bin inputs.bin out.bin\n", argv[0]); return 1; }
    FILE* fw = fopen(argv[1], "rb");
    if (!fw) { perror(argv[1]); return 1; }
    char magic[8]; int32_t layout = 0, count = 0;
    if (!rd(fw, magic, 8) || memcmp(magic, "MOCHAW01", 8) || !rd(fw, &layout, 1) || !rd(fw, &count, 1)) { fprintf(stderr, "bad weight file\n"); return 1; }

    mocha_cfg cfg = {};                               // configs/config.yaml:13-31
    cfg.T = 60; cfg.V = layout == 0 ? 24 : 22; cfg.C_in = 15; cfg.patch = 4; cfg.dim = 256;
    cfg.enc_depth = 2; cfg.enc_heads = 4; cfg.enc_dim_head = 128; cfg.enc_mlp = 512;
    cfg.dec_depth = 2; cfg.dec_heads = 4; cfg.dec_dim_head = 256; cfg.dec_mlp = 512;
    cfg.layout = layout;
    mocha_ctx* ctx = nullptr;
    if (mocha_create(&cfg, 0, &ctx)) { fprintf(stderr, "mocha_create: %s\n", mocha_last_error(nullptr)); return 3; }
    printf("library: %s, ABI %d, runtime %d\n", mocha_build_info(), mocha_abi_version(), mocha_runtime_version());

    for (int i = 0; i < count; ++i) {                 // load_state_dict (trainer.py:239-240)
        int32_t nl = 0, nd = 0;
        if (!rd(fw, &nl, 1) || nl <= 0 || nl > 256) { fprintf(stderr, "bad entry %d\n", i); return 1; }
        std::string name(nl, '\0');
        int64_t shape[8];
        if (!rd(fw, &name[0], (size_t)nl) || !rd(fw, &nd, 1) || nd < 0 || nd > 8 || !rd(fw, shape, (size_t)nd)) { fprintf(stderr, "bad entry %d\n", i); return 1; }
        size_t n = 1;
        for (int d = 0; d < nd; ++d) n *= (size_t)shape[d];
        std::vector<float> data(n);
        if (!rd(fw, data.data(), n)) { fprintf(stderr, "short data for %s\n", name.c_str()); return 1; }
        MOCHA_OK_(ctx, mocha_load_weight(ctx, name.c_str(), data.data(), shape, nd));
    }
    fclose(fw);
    MOCHA_OK_(ctx, mocha_finalize_weights(ctx));

    FILE* fi = fopen(argv[2], "rb");
    if (!fi) { perror(argv[2]); return 1; }
    int32_t B = 0, V = 0;
    if (!rd(fi, &B, 1) || !rd(fi, &V, 1) || V != cfg.V || B < 1 || B > 4096) { fprintf(stderr, "bad input header\n"); return 1; }
    const size_t nx = (size_t)B * 60 * V * 15, ng = 90 * 256, ne = (size_t)B * 90 * 256;
    std::vector<float> src(nx), cha(nx), mean(ng), sd(ng);
    if (!rd(fi, src.data(), nx) || !rd(fi, cha.data(), nx) || !rd(fi, mean.data(), ng) || !rd(fi, sd.data(), ng)) { fprintf(stderr, "short input file\n"); return 1; }
    fclose(fi);

    HIP_OK(hipSetDevice(0));
    hipStream_t s; HIP_OK(hipStreamCreate(&s));
    float *d_src, *d_cha, *d_mean, *d_sd, *d_Yf, *d_Yc, *d_enc, *d_nm; int32_t* d_idx;
    HIP_OK(hipMalloc(&d_src, nx * 4)); HIP_OK(hipMalloc(&d_cha, nx * 4)); HIP_OK(hipMalloc(&d_mean, ng * 4)); HIP_OK(hipMalloc(&d_sd, ng * 4));
    HIP_OK(hipMalloc(&d_Yf, nx * 4)); HIP_OK(hipMalloc(&d_Yc, nx * 4)); HIP_OK(hipMalloc(&d_enc, ne * 4)); HIP_OK(hipMalloc(&d_nm, ne * 4));
    HIP_OK(hipMalloc(&d_idx, (size_t)B * 4));
    HIP_OK(hipMemcpyAsync(d_src, src.data(), nx * 4, hipMemcpyHostToDevice, s)); HIP_OK(hipMemcpyAsync(d_cha, cha.data(), nx * 4, hipMemcpyHostToDevice, s));
    HIP_OK(hipMemcpyAsync(d_mean, mean.data(), ng * 4, hipMemcpyHostToDevice, s)); HIP_OK(hipMemcpyAsync(d_sd, sd.data(), ng * 4, hipMemcpyHostToDevice, s));

    // Generator.forward
    MOCHA_OK_(ctx, mocha_forward(ctx, d_src, d_cha, B, d_Yf, s));
    // the demo: the character clip's windows become the bank, the source windows are characterized against it
    MOCHA_OK_(ctx, mocha_encode(ctx, d_cha, B, d_enc, nullptr, d_mean, d_sd, d_nm, s));
    MOCHA_OK_(ctx, mocha_bank_set(ctx, d_nm, d_enc, B, 0, s));
    MOCHA_OK_(ctx, mocha_characterize(ctx, d_src, B, d_mean, d_sd, d_Yc, d_idx, s));

    std::vector<float> Yf(nx), Yc(nx); std::vector<int32_t> idx(B);
    HIP_OK(hipMemcpyAsync(Yf.data(), d_Yf, nx * 4, hipMemcpyDeviceToHost, s)); HIP_OK(hipMemcpyAsync(Yc.data(), d_Yc, nx * 4, hipMemcpyDeviceToHost, s));
    HIP_OK(hipMemcpyAsync(idx.data(), d_idx, (size_t)B * 4, hipMemcpyDeviceToHost, s));
    HIP_OK(hipStreamSynchronize(s));
    FILE* fo = fopen(argv[3], "wb");
    if (!fo) { perror(argv[3]); return 1; }
    fwrite(Yf.data(), 4, nx, fo); fwrite(Yc.data(), 4, nx, fo); fwrite(idx.data(), 4, (size_t)B, fo);
    fclose(fo);
    double a = 0; for (float v : Yc) a += v < 0 ? -v : v;
    printf("B = %d windows, V = %d: sum |Y_char| = %.6f, idx[0] = %d\n", B, V, a, idx[0]);
    mocha_destroy(ctx);
    return 0;
}
