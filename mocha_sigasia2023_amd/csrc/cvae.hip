// Small kernels of the CVAE sampler (model_CVAE.py): LayerNorm over 256 channels,
// prior token assembly with sin/cos positional encoding, latent selection / re-parameterisation and
// decoder memory + query initialisation.  The dense work runs on the shared GEMM / attention kernels.
#include "kernels.h"

namespace mocha {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// nn.LayerNorm(256): y = (x - mean) / sqrt(var + 1e-5) * w + b, biased variance; one wave per row
__global__ __launch_bounds__(256) void mocha_layernorm256(const float* __restrict__ x, const float* __restrict__ w,
                                                          const float* __restrict__ b, float* __restrict__ y, int rows) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    const f32x4 v = reinterpret_cast<const f32x4*>(x + (size_t)row * 256)[lane];
    float s = (v[0] + v[1]) + (v[2] + v[3]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    const float mean = s * (1.f / 256.f);
    const f32x4 d = v - mean;
    float q = (d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
    const float rstd = 1.f / sqrtf(q * (1.f / 256.f) + 1e-5f);
    const f32x4 wv = reinterpret_cast<const f32x4*>(w)[lane], bv = reinterpret_cast<const f32x4*>(b)[lane];
    reinterpret_cast<f32x4*>(y + (size_t)row * 256)[lane] = d * rstd * wv + bv;
}

hipError_t launch_layernorm256(const float* x, const float* w, const float* b, float* y, int rows, hipStream_t s) {
    if (rows <= 0) return hipSuccess;
    hipLaunchKernelGGL(mocha_layernorm256, dim3((rows + 3) / 4), dim3(256), 0, s, x, w, b, y, rows);
    return hipGetLastError();
}

// PriorNet.encode token assembly (model_CVAE.py:69-76): [mu_token, logvar_token, c] + pe[:2+nc]
__global__ __launch_bounds__(256) void mocha_cvae_prior_tokens(const float* __restrict__ c, const float* __restrict__ mu_tok,
                                                               const float* __restrict__ lv_tok, const float* __restrict__ pe,
                                                               float* __restrict__ out, int nc, int rows) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);        // (b, t) flattened; one wave per 256-channel row
    const int q = threadIdx.x & 63;
    if (row >= rows) return;
    const int ntok = nc + 2;
    const int b = row / ntok, t = row - b * ntok;
    f32x4 v;
    if (t == 0) v = reinterpret_cast<const f32x4*>(mu_tok)[q];
    else if (t == 1) v = reinterpret_cast<const f32x4*>(lv_tok)[q];
    else v = reinterpret_cast<const f32x4*>(c)[((size_t)b * nc + (t - 2)) * 64 + q];
    const f32x4 pv = reinterpret_cast<const f32x4*>(pe)[t * 64 + q];
    reinterpret_cast<f32x4*>(out)[(size_t)row * 64 + q] = v + pv;
}

hipError_t launch_cvae_prior_tokens(const float* c, const float* mu_tok, const float* lv_tok, const float* pe, float* out, int B, int nc,
                                    hipStream_t s) {
    if (B <= 0) return hipSuccess;
    const int rows = B * (nc + 2);
    hipLaunchKernelGGL(mocha_cvae_prior_tokens, dim3((rows + 3) / 4), dim3(256), 0, s, c, mu_tok, lv_tok, pe, out, nc, rows);
    return hipGetLastError();
}

// z = mu (+ eps * exp(0.5 logvar)), mu = x[:,0], logvar = x[:,1]  (model_CVAE.py:77-87);
// memory = [z, c] (:160); queries = pe[:nq] (zeros + positional encoding, :161-162)
// One wave per output row, 16 bytes per lane: row 0 of a clip is z, rows 1..nc its condition tokens, the next nq rows the queries
// (a first version walked all 271 rows in ONE workgroup per clip with 4-byte accesses: 35 us per frame of the CVAE branch for one clip).
__global__ __launch_bounds__(256) void mocha_cvae_latent(const float* __restrict__ x, int ntok, const float* __restrict__ eps,
                                                         const float* __restrict__ c, int nc, const float* __restrict__ pe, int nq,
                                                         float* __restrict__ mem, float* __restrict__ qout, float* __restrict__ mu_out,
                                                         float* __restrict__ lv_out, int rows_per_clip /*1 + nc + nq*/, int rows /*B * rows_per_clip*/) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), q = threadIdx.x & 63;
    if (row >= rows) return;
    const int b = row / rows_per_clip, r = row - b * rows_per_clip;
    if (r == 0) {
        const f32x4 mu = reinterpret_cast<const f32x4*>(x)[((size_t)b * ntok) * 64 + q], lv = reinterpret_cast<const f32x4*>(x)[((size_t)b * ntok + 1) * 64 + q];
        f32x4 z = mu;
        if (eps) {
            const f32x4 e = reinterpret_cast<const f32x4*>(eps)[(size_t)b * 64 + q];
            z[0] = mu[0] + e[0] * expf(0.5f * lv[0]); z[1] = mu[1] + e[1] * expf(0.5f * lv[1]);
            z[2] = mu[2] + e[2] * expf(0.5f * lv[2]); z[3] = mu[3] + e[3] * expf(0.5f * lv[3]);
        }
        if (mu_out) reinterpret_cast<f32x4*>(mu_out)[(size_t)b * 64 + q] = mu;
        if (lv_out) reinterpret_cast<f32x4*>(lv_out)[(size_t)b * 64 + q] = lv;
        reinterpret_cast<f32x4*>(mem)[((size_t)b * (nc + 1)) * 64 + q] = z;
    } else if (r <= nc) {
        reinterpret_cast<f32x4*>(mem)[((size_t)b * (nc + 1) + r) * 64 + q] = reinterpret_cast<const f32x4*>(c)[((size_t)b * nc + (r - 1)) * 64 + q];
    } else {
        const int i = r - 1 - nc;
        reinterpret_cast<f32x4*>(qout)[((size_t)b * nq + i) * 64 + q] = reinterpret_cast<const f32x4*>(pe)[(size_t)i * 64 + q];
    }
}

hipError_t launch_cvae_latent(const float* x, int ntok, const float* eps, const float* c, int nc, const float* pe, int nq, float* mem,
                              float* q, float* mu_out, float* logvar_out, int B, hipStream_t s) {
    if (B <= 0) return hipSuccess;
    const int rpc = 1 + nc + nq, rows = B * rpc;
    hipLaunchKernelGGL(mocha_cvae_latent, dim3((rows + 3) / 4), dim3(256), 0, s, x, ntok, eps, c, nc, pe, nq, mem, q, mu_out, logvar_out, rpc, rows);
    return hipGetLastError();
}

// Conditioning glue of the demo's CVAE branch (test_fullframework.py:446-449):
//   cond[b] = cat[(src_cnt[b] - src_mean)/src_std , (prev_cha[b] - cha_mean)/cha_std]   along tokens -> (B, 2n, 256)
__global__ __launch_bounds__(256) void mocha_cvae_condition(const float* __restrict__ src_cnt, const float* __restrict__ sm,
                                                            const float* __restrict__ ss, const float* __restrict__ prev,
                                                            const float* __restrict__ cm, const float* __restrict__ cs,
                                                            float* __restrict__ cond, int n, int rows /*B*2n*/) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), q = threadIdx.x & 63;
    if (row >= rows) return;
    const int b = row / (2 * n), t = row - b * 2 * n;
    const bool first = t < n;
    const int tt = first ? t : t - n;
    const f32x4 x = reinterpret_cast<const f32x4*>(first ? src_cnt : prev)[((size_t)b * n + tt) * 64 + q];
    const f32x4 m = reinterpret_cast<const f32x4*>(first ? sm : cm)[tt * 64 + q];
    const f32x4 sd = reinterpret_cast<const f32x4*>(first ? ss : cs)[tt * 64 + q];
    reinterpret_cast<f32x4*>(cond)[(size_t)row * 64 + q] = (x - m) / sd;
}

hipError_t launch_cvae_condition(const float* src_cnt, const float* sm, const float* ss, const float* prev, const float* cm,
                                 const float* cs, float* cond, int B, int n, hipStream_t s) {
    if (B <= 0) return hipSuccess;
    const int rows = B * 2 * n;
    hipLaunchKernelGGL(mocha_cvae_condition, dim3((rows + 3) / 4), dim3(256), 0, s, src_cnt, sm, ss, prev, cm, cs, cond, n, rows);
    return hipGetLastError();
}

// curr_cha_encoded = vae_output * cha_encoded_std + cha_encoded_mean   (test_fullframework.py:449)
__global__ __launch_bounds__(256) void mocha_scale_shift(const float* __restrict__ x, const float* __restrict__ mean,
                                                         const float* __restrict__ sd, float* __restrict__ out, int n, int rows) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), q = threadIdx.x & 63;
    if (row >= rows) return;
    const int t = row % n;
    const f32x4 v = reinterpret_cast<const f32x4*>(x)[(size_t)row * 64 + q];
    reinterpret_cast<f32x4*>(out)[(size_t)row * 64 + q] =
        v * reinterpret_cast<const f32x4*>(sd)[t * 64 + q] + reinterpret_cast<const f32x4*>(mean)[t * 64 + q];
}

hipError_t launch_scale_shift(const float* x, const float* mean, const float* sd, float* out, int B, int n, hipStream_t s) {
    if (B <= 0) return hipSuccess;
    const int rows = B * n;
    hipLaunchKernelGGL(mocha_scale_shift, dim3((rows + 3) / 4), dim3(256), 0, s, x, mean, sd, out, n, rows);
    return hipGetLastError();
}

}  // namespace mocha
