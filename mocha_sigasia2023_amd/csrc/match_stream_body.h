// The few-query bank scan's body (match_stream.hip), shared with the adaptive 1-byte first stage (match_scan8.hip).
#pragma once
#include "kernels.h"

namespace mocha {

typedef float ms_f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int ms_u32x4 __attribute__((ext_vector_type(4)));
typedef float ms_f32x2 __attribute__((ext_vector_type(2)));

static constexpr int MS_ROWS_PER_WAVE = 4;      // rows a wave carries through the D loop together
static constexpr int MS_WAVES = 4;
static constexpr int MS_CHUNK_F32 = 1280;       // elements of every query staged per step: 5 x 16 B per lane of fp32 bank ...
static constexpr int MS_CHUNK_BF16 = 1536;      // ... 3 x 16 B per lane of bf16 bank (whole 64-lane rounds; both divide 23 040)

__device__ __forceinline__ unsigned long long pack_key(float v, unsigned row) {
    unsigned u = __float_as_uint(v);
    u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
    return ((unsigned long long)u << 32) | row;
}

// all_keys != 0 (top-k queries): every row's key goes to partial[q][row] instead of one minimum per workgroup
// qs: Q * MS_CHUNK floats of LDS (16-byte aligned), wbest: MS_WAVES * Q words - the caller's (the kernels below declare them; the adaptive
// scan of match_scan8.hip shares one allocation between this body and its own)
template <int Q, bool BF16>
__device__ __forceinline__ void match_stream_body(const void* __restrict__ bank, const float* __restrict__ query, int nq, long long N, int D,
                                                  unsigned long long* __restrict__ partial /*[Q8][gridDim.x] or [Q8][N]*/,
                                                  int all_keys, unsigned long long* __restrict__ wgmin /*all_keys: [Q8][gridDim.x] too*/,
                                                  float* __restrict__ qs, unsigned long long (*wbest)[Q]) {
    constexpr int MS_CHUNK = BF16 ? MS_CHUNK_BF16 : MS_CHUNK_F32;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long long row0 = ((long long)blockIdx.x * MS_WAVES + wave) * MS_ROWS_PER_WAVE;

    // two partial sums per (row, query) so that the squares accumulate with packed fp32 math (v_pk_fma_f32): with
    // several queries the scan is VALU-bound, not HBM-bound
    ms_f32x2 acc[MS_ROWS_PER_WAVE][Q];
#pragma unroll
    for (int r = 0; r < MS_ROWS_PER_WAVE; ++r)
#pragma unroll
        for (int q = 0; q < Q; ++q) acc[r][q] = ms_f32x2{0.f, 0.f};

    const int nchunks = D / MS_CHUNK;
    for (int ch = 0; ch < nchunks; ++ch) {
        __syncthreads();
        // stage this chunk of every query: Q * 320 float4, 256 threads
        for (int i = tid; i < Q * (MS_CHUNK / 4); i += 256) {
            const int q = i / (MS_CHUNK / 4), o = i - q * (MS_CHUNK / 4);
            ms_f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (q < nq) v = reinterpret_cast<const ms_f32x4*>(query + (size_t)q * D + (size_t)ch * MS_CHUNK)[o];
            reinterpret_cast<ms_f32x4*>(qs)[i] = v;
        }
        __syncthreads();
        if (!BF16) {
            // 320 float4 per row-chunk: 5 per lane
            ms_f32x4 bv[MS_ROWS_PER_WAVE][5];
#pragma unroll
            for (int r = 0; r < MS_ROWS_PER_WAVE; ++r) {
                long long row = row0 + r;
                row = row < N ? row : N - 1;
                const ms_f32x4* bp = reinterpret_cast<const ms_f32x4*>(reinterpret_cast<const float*>(bank) + (size_t)row * D + (size_t)ch * MS_CHUNK);
#pragma unroll
                for (int i = 0; i < 5; ++i) bv[r][i] = __builtin_nontemporal_load(bp + lane + 64 * i);
            }
#pragma unroll
            for (int i = 0; i < 5; ++i)
#pragma unroll
                for (int q = 0; q < Q; ++q) {
                    const ms_f32x4 qv = reinterpret_cast<const ms_f32x4*>(qs)[q * (MS_CHUNK / 4) + lane + 64 * i];
#pragma unroll
                    for (int r = 0; r < MS_ROWS_PER_WAVE; ++r) {
                        const ms_f32x4 d = bv[r][i] - qv;
                        const ms_f32x2 dl = {d[0], d[1]}, dh = {d[2], d[3]};
                        acc[r][q] = __builtin_elementwise_fma(dl, dl, acc[r][q]);
                        acc[r][q] = __builtin_elementwise_fma(dh, dh, acc[r][q]);
                    }
                }
        } else {
            // bf16 bank: 1280 elements = 160 x 16 B per row-chunk: lanes 0..63 take pieces lane, lane+64, (lane+128 < 160)
            ms_u32x4 bv[MS_ROWS_PER_WAVE][3];
#pragma unroll
            for (int r = 0; r < MS_ROWS_PER_WAVE; ++r) {
                long long row = row0 + r;
                row = row < N ? row : N - 1;
                const ms_u32x4* bp = reinterpret_cast<const ms_u32x4*>(reinterpret_cast<const unsigned short*>(bank) + (size_t)row * D + (size_t)ch * MS_CHUNK);
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    const int piece = lane + 64 * i;
                    ms_u32x4 z = {0u, 0u, 0u, 0u};
                    bv[r][i] = piece < MS_CHUNK / 8 ? __builtin_nontemporal_load(bp + piece) : z;
                }
            }
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const int piece = lane + 64 * i;
                if (piece < MS_CHUNK / 8) {
#pragma unroll
                    for (int q = 0; q < Q; ++q) {
                        const ms_f32x4 q0 = reinterpret_cast<const ms_f32x4*>(qs)[q * (MS_CHUNK / 4) + piece * 2];
                        const ms_f32x4 q1 = reinterpret_cast<const ms_f32x4*>(qs)[q * (MS_CHUNK / 4) + piece * 2 + 1];
#pragma unroll
                        for (int r = 0; r < MS_ROWS_PER_WAVE; ++r) {
                            const ms_u32x4 w = bv[r][i];      // 8 bf16: element 2j in the low half of word j
                            ms_f32x2 d;
                            d = ms_f32x2{__uint_as_float(w[0] << 16), __uint_as_float(w[0] & 0xffff0000u)} - ms_f32x2{q0[0], q0[1]};
                            acc[r][q] = __builtin_elementwise_fma(d, d, acc[r][q]);
                            d = ms_f32x2{__uint_as_float(w[1] << 16), __uint_as_float(w[1] & 0xffff0000u)} - ms_f32x2{q0[2], q0[3]};
                            acc[r][q] = __builtin_elementwise_fma(d, d, acc[r][q]);
                            d = ms_f32x2{__uint_as_float(w[2] << 16), __uint_as_float(w[2] & 0xffff0000u)} - ms_f32x2{q1[0], q1[1]};
                            acc[r][q] = __builtin_elementwise_fma(d, d, acc[r][q]);
                            d = ms_f32x2{__uint_as_float(w[3] << 16), __uint_as_float(w[3] & 0xffff0000u)} - ms_f32x2{q1[2], q1[3]};
                            acc[r][q] = __builtin_elementwise_fma(d, d, acc[r][q]);
                        }
                    }
                }
            }
        }
    }
    // finish: one wave reduction per (row, query); the workgroup's minimum per query goes to partial[]
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
                const unsigned long long k = pack_key(v, (unsigned)row);      // v = squared distance
                kmin = k < kmin ? k : kmin;
                if (all_keys && lane == 0 && q < nq) partial[(size_t)q * N + row] = k;
            }
        }
        if (lane == 0) wbest[wave][q] = kmin;
    }
    __syncthreads();
    if (all_keys && !wgmin) return;
    if (tid < Q) {
        unsigned long long k = wbest[0][tid];
#pragma unroll
        for (int w = 1; w < MS_WAVES; ++w) k = wbest[w][tid] < k ? wbest[w][tid] : k;
        (all_keys ? wgmin : partial)[(size_t)tid * gridDim.x + blockIdx.x] = k;
    }
}


}  // namespace mocha
