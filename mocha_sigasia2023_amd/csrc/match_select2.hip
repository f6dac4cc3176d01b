// Many-query context matching, second stage (round 4): from the coarse pass's score slabs to the exact nearest row of every query
// (BallTree.query(k=1), test_fullframework.py:296,443).  Replaces mocha_match_select (match_mfma.hip, kept behind option "select2" = 0).
//
// mocha_match_select staged the whole query row in LDS (92 KB: one workgroup per CU), derived the error bound's norms from it, and
// re-evaluated every row inside the bound - with ONE bf16 query plane in the coarse pass that was ~8 rows for the unluckiest of 128
// queries, each a dependent memory round trip: 32 us for 128 x 4096, 95 us for 1024 x 4096 (rocprofv3: profiles/r04/c_select_trace.txt).
//
// Here the work that is usually unnecessary is not done:
//   * the producer of the centred queries (mocha_center_rows, or mocha_instnorm inside characterize) hands over the row statistics the
//     bound needs - ||q - c||^2 and the squared norm dq^2 of what the coarse pass's query planes leave out - so no pass over the query row;
//   * the bf16 coarse pass carries TWO query planes (16 significant bits): the bound 2 ||dq|| (||b_n - c|| + ||b_m - c||) + slack is
//     ~1/250 of the one-plane bound, and a query's candidate set is almost always the coarse minimum alone;
//   * a single candidate needs no evaluation when the caller wants the index only (characterize does): the kernel is then one read of
//     the query's scores and a reduction.  Otherwise the candidates' exact squared distances are evaluated in the direct form,
//     streamed - (q - c) - b per element against the centred bf16 rows the coarse pass scanned, or q - b on raw fp32 rows - by all
//     1024 threads per row, nothing staged in LDS.
// Bounds (match_mfma.hip): the coarse score v_n = ||b_n - c||^2 - 2 S_n differs from its exact value by e_n <= 2 ||dq|| ||b_n - c|| +
// rel (||q - c||^2 + ||b_n - c||^2); the true nearest row t satisfies v_t <= v_m + e_m + e_t for the coarse minimum m, so it is always
// among the candidates.  Every row of a query is summed in the same order: identical rows get identical distances, ties go to the
// lowest row.
#include "kernels.h"
#include "device_utils.h"

namespace mocha {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

static constexpr int S2_T = 1024, S2_W = S2_T / 64;
static constexpr int S2_KPT = 1;                     // score chunks (of 4 * S2_T rows) a thread keeps in registers
static constexpr int S2_CAP = 64;                    // candidate list capacity per pass
static constexpr int S2_NB = 3;                      // pieces of 8 elements per thread and batch of the row evaluation (all of D = 23040 in one batch)

__device__ __forceinline__ unsigned long long s2_key(float v, unsigned n) {            // order-preserving (value, index) key
    unsigned u = __float_as_uint(v);
    u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
    return ((unsigned long long)u << 32) | n;
}

struct Select2Params {
    const float* S; int ksplit; long long slab_stride; int lds;
    const float* bnorm; const float* query; const float* centre; const float* bank; const unsigned short* bank16;
    const float* qstat;           // per query: QSTAT_PARTS parts of ||q - c||^2, ||dq||^2
    long long N; int D; float margin_rel; int32_t* idx; float* dist;
};

template <bool B16>
__global__ __launch_bounds__(S2_T) void mocha_match_select2(Select2Params p) {
    __shared__ unsigned long long rk[S2_W];
    __shared__ float ra[S2_W];
    __shared__ unsigned long long r_key;
    __shared__ float r_d2;
    __shared__ int cand[S2_CAP], csort[S2_CAP];
    __shared__ int ncand;
    const int q = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long long N = p.N;
    const int D = p.D;
    if (tid == 0) ncand = 0;
    const float* qs = p.qstat + (size_t)q * 2 * QSTAT_PARTS;      // in flight beside the scores; the parts in order
    const float qn = ((qs[0] + qs[2]) + qs[4]) + qs[6], dq = ((qs[1] + qs[3]) + qs[5]) + qs[7];
    static_assert(QSTAT_PARTS == 4, "four parts");

    // ---- 1. coarse scores; thread t owns rows c0 + 4 t .. + 3 of every chunk (one 16-byte load per K slice and chunk)
    const float* Sq = p.S + (size_t)q * p.lds;
    const bool vec = ((p.lds | (int)(N & 3)) & 3) == 0 && ((size_t)p.slab_stride & 3) == 0;
    auto scores = [&](long long c0, float (&v)[4], float (&bn)[4]) __attribute__((always_inline)) {
        const long long n0 = c0 + tid * 4;
        float dot[4] = {0.f, 0.f, 0.f, 0.f};
        if (vec) {
            const long long nb = n0 + 3 < N ? n0 : (N - 4 > 0 ? N - 4 : 0);          // clamped: loads are unconditional
#pragma unroll 8
            for (int z = 0; z < p.ksplit; ++z) {
                const f32x4 d = *reinterpret_cast<const f32x4*>(Sq + (size_t)z * p.slab_stride + nb);
                dot[0] += d[0]; dot[1] += d[1]; dot[2] += d[2]; dot[3] += d[3];
            }
            const f32x4 b = *reinterpret_cast<const f32x4*>(p.bnorm + nb);
            bn[0] = b[0]; bn[1] = b[1]; bn[2] = b[2]; bn[3] = b[3];
        } else {
            long long nc[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) nc[e] = n0 + e < N ? n0 + e : N - 1;
#pragma unroll 4
            for (int z = 0; z < p.ksplit; ++z)
#pragma unroll
                for (int e = 0; e < 4; ++e) dot[e] += Sq[(size_t)z * p.slab_stride + nc[e]];
#pragma unroll
            for (int e = 0; e < 4; ++e) bn[e] = p.bnorm[nc[e]];
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float sc = bn[e] - 2.f * dot[e];
            v[e] = (n0 + e < N && sc == sc) ? sc : INFINITY;      // rows past the end and NaN scores never qualify
        }
    };
    float v[S2_KPT][4], bn[S2_KPT][4];
    unsigned long long key = ~0ull;
#pragma unroll
    for (int ch = 0; ch < S2_KPT; ++ch) {
        const long long c0 = (long long)ch * S2_T * 4;
        if (c0 < N) {                                            // uniform
            scores(c0, v[ch], bn[ch]);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const unsigned long long k = s2_key(v[ch][e], (unsigned)(c0 + tid * 4 + e));
                key = k < key ? k : key;
            }
        }
    }
    for (long long c0 = (long long)S2_KPT * S2_T * 4; c0 < N; c0 += S2_T * 4) {     // banks beyond 4 096 rows: not kept in registers
        float vv[4], bb[4];
        scores(c0, vv, bb);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const unsigned long long k = s2_key(vv[e], (unsigned)(c0 + tid * 4 + e));
            key = k < key ? k : key;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const unsigned long long k2 = __shfl_xor(key, o); key = k2 < key ? k2 : key; }
    if (lane == 0) rk[wave] = key;
    __syncthreads();
    if (wave == 0) {
        unsigned long long k = lane < S2_W ? rk[lane] : ~0ull;
#pragma unroll
        for (int o = S2_W / 2; o > 0; o >>= 1) { const unsigned long long k2 = __shfl_xor(k, o); k = k2 < k ? k2 : k; }
        if (lane == 0) r_key = k;
    }
    __syncthreads();
    const unsigned nmin = (unsigned)(r_key & 0xffffffffull);
    const unsigned umin = (unsigned)(r_key >> 32);
    const float vmin = __uint_as_float((umin & 0x80000000u) ? (umin & 0x7fffffffu) : ~umin);
    if (!(vmin < INFINITY)) {                                    // no finite score at all (NaN / inf inputs): row 0, distance NaN
        if (tid == 0) { p.idx[q] = 0; if (p.dist) p.dist[q] = __uint_as_float(0x7fc00000u); }
        return;
    }

    // ---- 2. candidates: the rows whose coarse score the bound cannot separate from the minimum's
    const float bmin = p.bnorm[nmin];
    const float base = 2.f * qn + bmin;
    const float dq2 = B16 ? 2.000002f * sqrtf(dq) : 0.f;
    const float sbmin = sqrtf(bmin);
    auto qualifies = [&](float sc, float b, unsigned n) -> bool {
        return n != nmin && sc < INFINITY && sc <= vmin + dq2 * (sqrtf(b) + sbmin) + p.margin_rel * (base + b);
    };
    auto list = [&](long long wlo, long long whi) __attribute__((always_inline)) {      // rows of [wlo, whi) that qualify -> cand[]
#pragma unroll
        for (int ch = 0; ch < S2_KPT; ++ch) {
            const long long c0 = (long long)ch * S2_T * 4;
            if (c0 < N) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const long long n = c0 + tid * 4 + e;
                    if (n >= wlo && n < whi && qualifies(v[ch][e], bn[ch][e], (unsigned)n)) {
                        const int pos = atomicAdd(&ncand, 1);
                        if (pos < S2_CAP) cand[pos] = (int)n;
                    }
                }
            }
        }
        for (long long c0 = (long long)S2_KPT * S2_T * 4; c0 < N; c0 += S2_T * 4) {
            if (c0 + S2_T * 4 <= wlo || c0 >= whi) continue;     // uniform
            float vv[4], bb[4];
            scores(c0, vv, bb);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const long long n = c0 + tid * 4 + e;
                if (n >= wlo && n < whi && qualifies(vv[e], bb[e], (unsigned)n)) {
                    const int pos = atomicAdd(&ncand, 1);
                    if (pos < S2_CAP) cand[pos] = (int)n;
                }
            }
        }
    };
    list(0, N);
    __syncthreads();
    const int total = ncand;
    if (total == 0 && !p.dist) {                                 // the minimum stands alone and only the index is wanted: done
        if (tid == 0) p.idx[q] = (int)nmin;
        return;
    }

    // ---- 3. exact squared distance of one row: pieces of 8 elements, thread t takes pieces t, t + 1024, ...; all loads of a batch are issued
    // before any is used.  16-byte buffer loads: base addresses in SGPRs, one 32-bit offset per piece.
    const int np = D / 8;
    const __amdgpu_buffer_rsrc_t rsq = make_rsrc(p.query + (size_t)q * D), rsc = make_rsrc(p.centre);
    auto exact = [&](long long row_) -> float {
        const long long row = (long long)__builtin_amdgcn_readfirstlane((int)row_);       // uniform: the row's base goes to SGPRs
        const __amdgpu_buffer_rsrc_t rsb = B16 ? make_rsrc(p.bank16 + (size_t)row * D) : make_rsrc(p.bank + (size_t)row * D);
        float a = 0.f;
        for (int it0 = 0; it0 * S2_T < np; it0 += S2_NB) {
            f32x4 q0[S2_NB], q1[S2_NB], c0[B16 ? S2_NB : 1], c1[B16 ? S2_NB : 1], b0[B16 ? 1 : S2_NB], b1[B16 ? 1 : S2_NB];
            u32x4 bw[B16 ? S2_NB : 1];
#pragma unroll
            for (int u = 0; u < S2_NB; ++u) {
                int pc = (it0 + u) * S2_T + tid;
                pc = pc < np ? pc : np - 1;
                q0[u] = bload(rsq, (unsigned)pc * 32u, 0); q1[u] = bload(rsq, (unsigned)pc * 32u, 16);
                if (B16) {
                    c0[u] = bload(rsc, (unsigned)pc * 32u, 0); c1[u] = bload(rsc, (unsigned)pc * 32u, 16);
                    bw[u] = __builtin_bit_cast(u32x4, bload(rsb, (unsigned)pc * 16u, 0));
                } else { b0[u] = bload(rsb, (unsigned)pc * 32u, 0); b1[u] = bload(rsb, (unsigned)pc * 32u, 16); }
            }
#pragma unroll
            for (int u = 0; u < S2_NB; ++u) {
                if ((it0 + u) * S2_T + tid < np) {
                    f32x4 x0 = q0[u], x1 = q1[u], y0, y1;
                    if (B16) {
                        x0 = q0[u] - c0[u]; x1 = q1[u] - c1[u];      // the subtraction mocha_center_rows / mocha_instnorm made
                        const u32x4 w = bw[u];
                        y0 = f32x4{__uint_as_float(w[0] << 16), __uint_as_float(w[0] & 0xffff0000u), __uint_as_float(w[1] << 16), __uint_as_float(w[1] & 0xffff0000u)};
                        y1 = f32x4{__uint_as_float(w[2] << 16), __uint_as_float(w[2] & 0xffff0000u), __uint_as_float(w[3] << 16), __uint_as_float(w[3] & 0xffff0000u)};
                    } else { y0 = b0[u]; y1 = b1[u]; }
                    const f32x4 d0 = x0 - y0, d1 = x1 - y1;
                    a = fmaf(d0[0], d0[0], a); a = fmaf(d0[1], d0[1], a); a = fmaf(d0[2], d0[2], a); a = fmaf(d0[3], d0[3], a);
                    a = fmaf(d1[0], d1[0], a); a = fmaf(d1[1], d1[1], a); a = fmaf(d1[2], d1[2], a); a = fmaf(d1[3], d1[3], a);
                }
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o);
        __syncthreads();                                         // ra may still be read from the previous row
        if (lane == 0) ra[wave] = a;
        __syncthreads();
        if (tid == 0) {
            float s = ra[0];
            for (int w = 1; w < S2_W; ++w) s += ra[w];           // fixed order
            r_d2 = s;
        }
        __syncthreads();
        return r_d2;
    };

    // (exact distance bits, row): distances are >= 0 or NaN (sorts last)
    unsigned long long best = ((unsigned long long)__float_as_uint(exact((long long)nmin)) << 32) | nmin;
    const bool windowed = total > S2_CAP;                        // uniform
    const long long nwin = windowed ? (N + S2_CAP - 1) / S2_CAP : 1;
    for (long long w = 0; w < nwin && total > 0; ++w) {
        int nc = total;
        if (windowed) {                                          // rare: rebuild the list for rows [w CAP, (w + 1) CAP): at most CAP rows
            __syncthreads();
            if (tid == 0) ncand = 0;
            __syncthreads();
            list(w * S2_CAP, (w + 1) * S2_CAP);
            __syncthreads();
            nc = ncand;
            if (nc == 0) continue;                               // uniform
        }
        // rank sort by row index: the evaluation order does not depend on the atomics' order
        if (tid < nc) {
            const int mine = cand[tid];
            int r = 0;
            for (int i = 0; i < nc; ++i) r += cand[i] < mine;
            csort[r] = mine;
        }
        __syncthreads();
        for (int i = 0; i < nc; ++i) {
            const int row = csort[i];
            const unsigned long long k = ((unsigned long long)__float_as_uint(exact(row)) << 32) | (unsigned)row;
            best = k < best ? k : best;
        }
    }
    if (tid == 0) {
        p.idx[q] = (int)(best & 0xffffffffull);
        if (p.dist) p.dist[q] = sqrtf(__uint_as_float((unsigned)(best >> 32)));
    }
}

hipError_t launch_match_select2(const float* S, int ksplit, long long slab_stride, int lds, const float* bnorm, const float* query,
                                const float* centre, const float* bank, const void* bank16, const float* qstat, float margin_rel, int Q,
                                int64_t N, int D, int32_t* idx, float* dist, hipStream_t s) {
    if (Q <= 0) return hipSuccess;
    if (D % 256 || D > (1 << 24) || N < 1 || N > 0x7ffffff0ll || !(margin_rel >= 0.f) || !qstat) return hipErrorInvalidValue;
    Select2Params p;
    p.S = S; p.ksplit = ksplit; p.slab_stride = slab_stride; p.lds = lds; p.bnorm = bnorm; p.query = query; p.centre = centre;
    p.bank = bank; p.bank16 = (const unsigned short*)bank16; p.qstat = qstat; p.N = (long long)N; p.D = D; p.margin_rel = margin_rel;
    p.idx = idx; p.dist = dist;
    if (bank16) hipLaunchKernelGGL(mocha_match_select2<true>, dim3((unsigned)Q), dim3(S2_T), 0, s, p);
    else hipLaunchKernelGGL(mocha_match_select2<false>, dim3((unsigned)Q), dim3(S2_T), 0, s, p);
    return hipGetLastError();
}

}  // namespace mocha
