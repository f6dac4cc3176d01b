// A ONE-BYTE first stage for the few-query scan of a large fp32 bank (round 6; option "scan8", default off).
//
// The few-query matcher is HBM-bound: one streamed query against 16 384 rows reads the bank's centred bf16 copy, 2 B per value
// (755 MB, ~130 us).  mocha_bank_set can also keep the centred rows as biased bytes with a per-row scale,
//     b8_n = s_n (u8_n - 128),   s_n = max_i |b_ni - c_i| / 127,   u8 = 128 + rint((b - c) / s_n),
// and per row  rho8_n >= || (b_n - c) - b8_n ||  (the quantisation residual's norm, measured, plus 1e-6 ||b_n - c||).  The scan then reads
// 1 B per value and produces every row's coarse distance d8_n = ||(q - c) - b8_n|| in the direct form; by the triangle inequality
// | ||q - b_n|| - d8_n | <= rho8_n, so mocha_match_refine (match_stream.hip) - unchanged: it takes any coarse keys with their residual
// bounds - re-evaluates on the fp32 rows exactly the rows the byte image cannot exclude and returns the result of the exact fp32 search.
//
// Whether that pays depends on the GEOMETRY: a query that sits close to a few rows of the bank (gap to the rest much larger than
// 2 rho8 ~ 3 on unit-variance rows) leaves a handful of candidates; on a bank of independent N(0, 1) rows every distance is 214.7 +- 1
// and thousands of rows stay inside the bound - the byte scan is then wasted.  So the stage is ADAPTIVE, on the device (the calls
// may sit in a captured graph): the refine kernel counts its candidates and, if one of its 64 slices holds more than S8_DEGENERATE (12) of them, sets a sticky
// mode word; from the next call on the same launch runs the bf16 scan of round 3 instead (one kernel, branch at its top: no extra
// launches either way).  mocha_bank_set clears the word.
#include "kernels.h"
#include "match_stream_body.h"

namespace mocha {

static constexpr int MS_CHUNK_I8 = 2560;        // elements of every query staged per step: 160 pieces of 16 bytes per row (lanes take pieces lane, lane + 64, lane + 128 < 160)

// rows (x - centre) -> biased bytes + scale + residual bound; one workgroup per row
__global__ __launch_bounds__(256) void mocha_to_i8(const float* __restrict__ x, const float* __restrict__ centre, unsigned char* __restrict__ y,
                                                   float* __restrict__ scale, float* __restrict__ rho, int cols) {
    __shared__ float red[3][4];
    __shared__ float s_inv, s_s;
    const size_t row = blockIdx.x;
    const int tid = threadIdx.x;
    float mx = 0.f;
    for (int i = tid; i < cols; i += 256) mx = fmaxf(mx, fabsf(x[row * cols + i] - centre[i]));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    if ((tid & 63) == 0) red[0][tid >> 6] = mx;
    __syncthreads();
    if (tid == 0) {
        const float m = fmaxf(fmaxf(red[0][0], red[0][1]), fmaxf(red[0][2], red[0][3]));
        const float s = m > 0.f && m < __builtin_inff() ? m / 127.f : 1.f;
        s_s = s; s_inv = 1.f / s;
    }
    __syncthreads();
    const float s = s_s, inv = s_inv;
    float a = 0.f, nn = 0.f;
    for (int i = tid; i < cols; i += 256) {
        const float t = x[row * cols + i] - centre[i];
        float q = rintf(t * inv);
        q = q == q ? fminf(fmaxf(q, -127.f), 127.f) : 0.f;
        y[row * cols + i] = (unsigned char)(int)(q + 128.f);
        const float d = t - s * q;              // s * q: ONE rounding - the value the scan rebuilds as fma(u8, s, -128 s)
        a = fmaf(d, d, a); nn = fmaf(t, t, nn);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { a += __shfl_xor(a, o); nn += __shfl_xor(nn, o); }
    if ((tid & 63) == 0) { red[1][tid >> 6] = a; red[2][tid >> 6] = nn; }
    __syncthreads();
    if (tid == 0) {
        const float r2 = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]), n2 = (red[2][0] + red[2][1]) + (red[2][2] + red[2][3]);
        scale[row] = s;
        rho[row] = sqrtf(r2) * 1.000001f + 1e-6f * sqrtf(n2);
    }
}

hipError_t launch_to_i8(const float* x, const float* centre, void* y, float* scale, float* rho, int64_t rows, int cols, hipStream_t s) {
    if (rows <= 0) return hipSuccess;
    hipLaunchKernelGGL(mocha_to_i8, dim3((unsigned)rows), dim3(256), 0, s, x, centre, (unsigned char*)y, scale, rho, cols);
    return hipGetLastError();
}

// every row's coarse key from the byte image (mode 0) or from the bf16 copy (mode 1: the round-3 scan, same body as mocha_match_stream)
template <int Q>
__global__ __launch_bounds__(256) void mocha_match_scan_adaptive(const unsigned char* __restrict__ bank8, const float* __restrict__ scale,
                                                                 const void* __restrict__ bank16, const float* __restrict__ query /*centred*/,
                                                                 int nq, long long N, int D, unsigned long long* __restrict__ keys /*[8][N]*/,
                                                                 unsigned long long* __restrict__ wgmin /*[8][gridDim.x]*/, unsigned* __restrict__ mode /*{cur, next}*/) {
    extern __shared__ __attribute__((aligned(16))) float s8_sm[];      // Q * MS_CHUNK_I8 floats, then MS_WAVES * Q words
    float* qs = s8_sm;
    unsigned long long (*wbest)[Q] = reinterpret_cast<unsigned long long (*)[Q]>(s8_sm + Q * MS_CHUNK_I8);
    // the mode of THIS call: `next` is stable while the scan runs (the previous call's refine has finished: same stream); one thread
    // publishes it as `cur` for this call's refine, which may already be writing `next` again while its other workgroups start
    const unsigned m = mode[1];
    if (blockIdx.x == 0 && threadIdx.x == 0) mode[0] = m;
    if (m) { match_stream_body<Q, true>(bank16, query, nq, N, D, keys, 1, wgmin, qs, wbest); return; }

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long long row0 = ((long long)blockIdx.x * MS_WAVES + wave) * MS_ROWS_PER_WAVE;
    ms_f32x2 acc[MS_ROWS_PER_WAVE][Q];
    ms_f32x2 sc[MS_ROWS_PER_WAVE], bi[MS_ROWS_PER_WAVE];
#pragma unroll
    for (int r = 0; r < MS_ROWS_PER_WAVE; ++r) {
        long long row = row0 + r; row = row < N ? row : N - 1;
        const float s = scale[row];
        sc[r] = ms_f32x2{s, s}; bi[r] = ms_f32x2{-128.f * s, -128.f * s};
#pragma unroll
        for (int q = 0; q < Q; ++q) acc[r][q] = ms_f32x2{0.f, 0.f};
    }
    const int nchunks = D / MS_CHUNK_I8;
    for (int ch = 0; ch < nchunks; ++ch) {
        __syncthreads();
        for (int i = tid; i < Q * (MS_CHUNK_I8 / 4); i += 256) {
            const int q = i / (MS_CHUNK_I8 / 4), o = i - q * (MS_CHUNK_I8 / 4);
            ms_f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (q < nq) v = reinterpret_cast<const ms_f32x4*>(query + (size_t)q * D + (size_t)ch * MS_CHUNK_I8)[o];
            reinterpret_cast<ms_f32x4*>(qs)[i] = v;
        }
        __syncthreads();
        ms_u32x4 bv[MS_ROWS_PER_WAVE][3];
#pragma unroll
        for (int r = 0; r < MS_ROWS_PER_WAVE; ++r) {
            long long row = row0 + r; row = row < N ? row : N - 1;
            const ms_u32x4* bp = reinterpret_cast<const ms_u32x4*>(bank8 + (size_t)row * D + (size_t)ch * MS_CHUNK_I8);
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const int piece = lane + 64 * i;
                const ms_u32x4 z = {0u, 0u, 0u, 0u};
                bv[r][i] = piece < MS_CHUNK_I8 / 16 ? __builtin_nontemporal_load(bp + piece) : z;
            }
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int piece = lane + 64 * i;
            if (piece < MS_CHUNK_I8 / 16) {
#pragma unroll
                for (int w = 0; w < 4; ++w) {                       // dword w of the piece: elements 4 w .. 4 w + 3
#pragma unroll
                    for (int q = 0; q < Q; ++q) {
                        const ms_f32x4 qv = reinterpret_cast<const ms_f32x4*>(qs)[q * (MS_CHUNK_I8 / 4) + piece * 4 + w];
#pragma unroll
                        for (int r = 0; r < MS_ROWS_PER_WAVE; ++r) {
                            const unsigned u = bv[r][i][w];
                            const ms_f32x2 lo = {(float)(u & 0xffu), (float)((u >> 8) & 0xffu)}, hi = {(float)((u >> 16) & 0xffu), (float)(u >> 24)};      // v_cvt_f32_ubyte0 .. 3
                            ms_f32x2 d = __builtin_elementwise_fma(lo, sc[r], bi[r]) - ms_f32x2{qv[0], qv[1]};
                            acc[r][q] = __builtin_elementwise_fma(d, d, acc[r][q]);
                            d = __builtin_elementwise_fma(hi, sc[r], bi[r]) - ms_f32x2{qv[2], qv[3]};
                            acc[r][q] = __builtin_elementwise_fma(d, d, acc[r][q]);
                        }
                    }
                }
            }
        }
    }
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        unsigned long long kmin = ~0ull;
#pragma unroll
        for (int r = 0; r < MS_ROWS_PER_WAVE; ++r) {
            const long long row = row0 + r;
            float v = acc[r][q][0] + acc[r][q][1];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
            if (row < N) {
                const unsigned long long k = pack_key(v, (unsigned)row);
                kmin = k < kmin ? k : kmin;
                if (lane == 0 && q < nq) keys[(size_t)q * N + row] = k;
            }
        }
        if (lane == 0) wbest[wave][q] = kmin;
    }
    __syncthreads();
    if (tid < Q) {
        unsigned long long k = wbest[0][tid];
#pragma unroll
        for (int w = 1; w < MS_WAVES; ++w) k = wbest[w][tid] < k ? wbest[w][tid] : k;
        wgmin[(size_t)tid * gridDim.x + blockIdx.x] = k;
    }
}

template <int Q> static constexpr size_t s8_lds() { return (size_t)Q * MS_CHUNK_I8 * 4 + (size_t)MS_WAVES * Q * 8; }

hipError_t match_scan8_init() {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&mocha_match_scan_adaptive<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)s8_lds<1>());
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&mocha_match_scan_adaptive<2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)s8_lds<2>());
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&mocha_match_scan_adaptive<4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)s8_lds<4>());
    return e;
}

// Q <= 4 queries (one launch pair): scratch as for launch_match_scan16 (+ the two mode words behind its head: match_scan8_mode_word())
hipError_t launch_match_scan8(const void* bank8, const float* scale8, const float* rho8, const void* bank16, const float* rho16, const float* bank,
                              const float* qc, const float* query, int Q, int64_t N, int D, unsigned long long* scratch, int32_t* idx, float* dist,
                              hipStream_t s) {
    if (Q <= 0) return hipSuccess;
    if (Q > 4 || D % MS_CHUNK_I8 != 0 || D % MS_CHUNK_BF16 != 0 || D % 256 || D > 23040 || N < 1 || N > 0x7ffffff0ll) return hipErrorInvalidValue;
    unsigned long long* keys = scratch + match_scan16_scratch_head_words();
    const unsigned grid = (unsigned)((N + MS_WAVES * MS_ROWS_PER_WAVE - 1) / (MS_WAVES * MS_ROWS_PER_WAVE));
    unsigned long long* wgmin = keys + (size_t)8 * N;
    unsigned* mode = reinterpret_cast<unsigned*>(scratch + match_scan8_mode_word());
#define S8_LAUNCH(QQ) hipLaunchKernelGGL((mocha_match_scan_adaptive<QQ>), dim3(grid), dim3(256), s8_lds<QQ>(), s, (const unsigned char*)bank8, scale8, bank16, qc, Q, (long long)N, D, keys, wgmin, mode)
    if (Q == 1) S8_LAUNCH(1);
    else if (Q == 2) S8_LAUNCH(2);
    else S8_LAUNCH(4);
#undef S8_LAUNCH
    return launch_match_refine(keys, wgmin, (int)grid, rho16, rho8, mode, bank, query, Q, N, D, idx, dist, scratch, s);
}

}  // namespace mocha
