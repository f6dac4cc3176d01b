// Context matching for FEW queries against a LARGE bank: the HBM-bound regime (streaming one query
// per frame against a 4k-16k entry bank, BASELINE configs[4]; SURVEY.md §8d "bank scan").
//
// exact 1-NN = argmin_b sum_d (q_d - b_d)^2          (test_fullframework.py:296,443)
// in the DIRECT form: the scan is HBM-bound, so the subtraction is free, and it has none of the cancellation of
// ||b||^2 - 2 q.b (feature rows of one character sit close together far from the origin: ||b||^2 ~ 1e5, gaps between the
// best candidates ~ 1e-2).
// The bank is read exactly once per launch, 16 bytes per lane per load, straight to registers
// (no LDS round trip for streamed-once data); the <= 8 query vectors of a launch are staged through
// LDS one D-chunk at a time and reused by every bank row of the workgroup.  Each wave owns whole
// rows (lanes stride the 23 040-dim row), so a row's dot products are finished by one wave
// reduction.  Candidates are compared as 64-bit keys (order-preserving bits of the value << 32 | row),
// which makes ties go to the lowest index; each workgroup writes one partial minimum per query and
// the finish kernel reduces them (no atomics: 16k same-address atomics cost more than the scan).
// The bank may be fp32 or bf16 (widened exactly on load; half the HBM bytes).
#include "kernels.h"
#include "match_stream_body.h"

namespace mocha {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// all_keys != 0 (top-k queries): every row's key goes to partial[q][row] instead of one minimum per workgroup
template <int Q, bool BF16>
__global__ __launch_bounds__(256) void mocha_match_stream(const void* __restrict__ bank,
                                                          const float* __restrict__ query, int nq, long long N, int D,
                                                          unsigned long long* __restrict__ partial /*[Q8][gridDim.x] or [Q8][N]*/,
                                                          int all_keys, unsigned long long* __restrict__ wgmin = nullptr /*all_keys: [Q8][gridDim.x] too*/) {
    constexpr int MS_CHUNK = BF16 ? MS_CHUNK_BF16 : MS_CHUNK_F32;
    __shared__ __attribute__((aligned(16))) float qs[Q * MS_CHUNK];
    __shared__ unsigned long long wbest[MS_WAVES][Q];
    match_stream_body<Q, BF16>(bank, query, nq, N, D, partial, all_keys, wgmin, qs, wbest);
}

// partial minima -> idx[q]; Euclidean distance to the winner in the direct (q-b)^2 form
template <bool BF16>
__global__ __launch_bounds__(256) void mocha_match_finish(const unsigned long long* __restrict__ partial, int nwg,
                                                          const void* __restrict__ bank, const float* __restrict__ query, int D,
                                                          int32_t* __restrict__ idx, float* __restrict__ dist) {
    __shared__ unsigned long long kred[256];
    __shared__ float red[4];
    const int q = blockIdx.x, tid = threadIdx.x;
    const unsigned long long* pq = partial + (size_t)q * nwg;          // query-major: partial[q][wg]
    unsigned long long k = ~0ull;
    for (int i = tid; i < nwg; i += 256) k = pq[i] < k ? pq[i] : k;
    kred[tid] = k;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) kred[tid] = kred[tid + o] < kred[tid] ? kred[tid + o] : kred[tid];
        __syncthreads();
    }
    const unsigned row = (unsigned)(kred[0] & 0xffffffffull);
    if (tid == 0) idx[q] = (int32_t)row;
    if (!dist) return;
    float a = 0.f;
    for (int i = tid; i < D; i += 256) {
        float b;
        if (BF16) b = __uint_as_float((unsigned)reinterpret_cast<const unsigned short*>(bank)[(size_t)row * D + i] << 16);
        else b = reinterpret_cast<const float*>(bank)[(size_t)row * D + i];
        const float d = query[(size_t)q * D + i] - b;
        a = fmaf(d, d, a);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o);
    if ((tid & 63) == 0) red[tid >> 6] = a;
    __syncthreads();
    if (tid == 0) dist[q] = sqrtf((red[0] + red[1]) + (red[2] + red[3]));
}

size_t match_stream_scratch(int Q, int64_t N) {        // u64 words of partial[] needed
    const int rows_per_wg = MS_WAVES * MS_ROWS_PER_WAVE;
    const size_t nwg = (size_t)((N + rows_per_wg - 1) / rows_per_wg);
    return (size_t)((Q + 7) / 8) * 8 * nwg;
}

hipError_t launch_match_stream(const void* bank, int bank_bf16, const float* query, int Q, int64_t N, int D,
                               unsigned long long* partial, int32_t* idx, float* dist, hipStream_t s) {
    if (Q <= 0) return hipSuccess;
    if (D % (bank_bf16 ? MS_CHUNK_BF16 : MS_CHUNK_F32) != 0) return hipErrorInvalidValue;
    const int rows_per_wg = MS_WAVES * MS_ROWS_PER_WAVE;
    const unsigned grid = (unsigned)((N + rows_per_wg - 1) / rows_per_wg);
    for (int q0 = 0; q0 < Q; q0 += 8) {                 // more than 8 queries: one bank pass per 8
        const int nq = (Q - q0) < 8 ? (Q - q0) : 8;
        const float* qp = query + (size_t)q0 * D;
        unsigned long long* pp = partial + (size_t)q0 * grid;      // partial[q][wg], q global
#define MS_LAUNCH(QQ)                                                                                          \
        do {                                                                                                   \
            if (bank_bf16) hipLaunchKernelGGL((mocha_match_stream<QQ, true>), dim3(grid), dim3(256), 0, s, bank, qp, nq, (long long)N, D, pp, 0); \
            else hipLaunchKernelGGL((mocha_match_stream<QQ, false>), dim3(grid), dim3(256), 0, s, bank, qp, nq, (long long)N, D, pp, 0);          \
        } while (0)
        if (nq == 1) MS_LAUNCH(1);
        else if (nq == 2) MS_LAUNCH(2);
        else if (nq <= 4) MS_LAUNCH(4);
        else MS_LAUNCH(8);
#undef MS_LAUNCH
    }
    if (bank_bf16) hipLaunchKernelGGL((mocha_match_finish<true>), dim3(Q), dim3(256), 0, s, partial, (int)grid, bank, query, D, idx, dist);
    else hipLaunchKernelGGL((mocha_match_finish<false>), dim3(Q), dim3(256), 0, s, partial, (int)grid, bank, query, D, idx, dist);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------------
// k nearest rows (the tree.query(k > 1) of scikit-learn's BallTree; the reference itself only uses k = 1): the scan above
// writes the exact squared distance of EVERY row as a (value, row) key; one workgroup per query then takes the k smallest
// keys in k passes - keys are unique, so pass r selects the smallest key greater than pass r - 1's, no marking needed.
// Distances ascending, ties to the lowest row.  One bank scan per 8 queries: a convenience, not a fast path.
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void mocha_match_topk_finish(const unsigned long long* __restrict__ keys /*[Q][N]*/, long long N, int k,
                                                               int32_t* __restrict__ idx /*[Q][k]*/, float* __restrict__ dist /*[Q][k] or null*/) {
    __shared__ unsigned long long kred[256];
    const int q = blockIdx.x, tid = threadIdx.x;
    const unsigned long long* kq = keys + (size_t)q * N;
    unsigned long long prev = 0;
    for (int r = 0; r < k; ++r) {
        unsigned long long best = ~0ull;
        for (long long i = tid; i < N; i += 256) {
            const unsigned long long v = kq[i];
            if ((r == 0 || v > prev) && v < best) best = v;
        }
        kred[tid] = best;
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) {
            if (tid < o) kred[tid] = kred[tid + o] < kred[tid] ? kred[tid + o] : kred[tid];
            __syncthreads();
        }
        prev = kred[0];
        __syncthreads();
        if (tid == 0) {
            if (prev == ~0ull) { idx[(size_t)q * k + r] = -1; if (dist) dist[(size_t)q * k + r] = INFINITY; }      // fewer than k rows
            else {
                idx[(size_t)q * k + r] = (int32_t)(prev & 0xffffffffull);
                unsigned u = (unsigned)(prev >> 32);
                u = (u & 0x80000000u) ? (u & 0x7fffffffu) : ~u;
                if (dist) dist[(size_t)q * k + r] = sqrtf(__uint_as_float(u));
            }
        }
    }
}

hipError_t launch_match_topk(const void* bank, int bank_bf16, const float* query, int Q, int64_t N, int D, int k,
                             unsigned long long* keys /*8 * N words*/, int32_t* idx, float* dist, hipStream_t s) {
    if (Q <= 0) return hipSuccess;
    if (k < 1 || D % (bank_bf16 ? MS_CHUNK_BF16 : MS_CHUNK_F32) != 0) return hipErrorInvalidValue;
    const int rows_per_wg = MS_WAVES * MS_ROWS_PER_WAVE;
    const unsigned grid = (unsigned)((N + rows_per_wg - 1) / rows_per_wg);
    for (int q0 = 0; q0 < Q; q0 += 8) {                 // one bank pass per 8 queries; the key buffer is reused (same stream)
        const int nq = (Q - q0) < 8 ? (Q - q0) : 8;
        const float* qp = query + (size_t)q0 * D;
        if (bank_bf16) hipLaunchKernelGGL((mocha_match_stream<8, true>), dim3(grid), dim3(256), 0, s, bank, qp, nq, (long long)N, D, keys, 1);
        else hipLaunchKernelGGL((mocha_match_stream<8, false>), dim3(grid), dim3(256), 0, s, bank, qp, nq, (long long)N, D, keys, 1);
        hipLaunchKernelGGL(mocha_match_topk_finish, dim3(nq), dim3(256), 0, s, keys, (long long)N, k, idx + (size_t)q0 * k,
                           dist ? dist + (size_t)q0 * k : nullptr);
    }
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------------
// fp32 bank at HALF the scan bytes, same result (round 3).  The scan is HBM-bound, so an fp32 bank of N rows costs N * 92 KB
// per pass.  mocha_bank_set keeps, beside the fp32 rows, their centred bf16 copy b16_n = bf16(b_n - c) and per row
//     rho_n >= || (b_n - c) - b16_n ||                  (mocha_rowresid: the rounding residual's norm, plus 1e-6 ||b_n - c||)
// The few-query matcher then (1) scans the bf16 copy for EVERY row's coarse distance d16_n = ||(q - c) - b16_n|| (the kernel above
// in all_keys mode: half the bytes), and (2) mocha_match_refine re-evaluates exactly - direct sum (q - b_n)^2 over the fp32 row -
// every row the rounding cannot exclude.  By the triangle inequality | ||q - b_n|| - d16_n | <= rho_n (+ the fp32 accumulation of the
// scan, priced at 2e-5 d16_n), so the true nearest row n* satisfies
//     d16_n* - rho_n* <= ||q - b_n*|| <= ||q - b_m|| <= d16_m + rho_m      for the coarse minimum m,
// and every row with  d16_n - rho'_n <= d16_m + rho'_m  (rho'_n = rho_n + 2e-5 d16_n) is a candidate.  The smallest exact distance wins,
// ties to the lowest row: the result of the exact fp32 search, indices and distance.  On N(0,1) banks 2-4 rows qualify; a bank of
// identical rows makes every row a candidate (index windows of RF_CAP rows, all evaluated).
// RF_SPLIT 1024-thread workgroups per query, each re-ranking the candidates of its slice of the rows (a bank whose rows crowd within
// the copy's rounding of the best one - hundreds of candidates - is spread over the chip); the workgroup that finishes last picks the
// smallest (distance, row) key.  The coarse minimum comes from the scan's per-workgroup minima (the scan writes them beside the keys), so
// a workgroup reads its own slice of the keys only (RF_SPLIT = 64 slices since round 6: a clip whose stride-1 neighbours all sit in the bank leaves ~300
// candidates per query - at 16 slices their re-rank ran on 16 CUs; same box, streamed step on such a bank 0.438 -> 0.408 ms, an uncrowded call + 1.5 us:
// profiles/r06/h_refine_slices_ab.txt).  A row's terms are summed in 16 fixed segments, one per wave, added in segment
// order: identical rows get identical distances whatever else is in the list and however the slices fall.
// ---------------------------------------------------------------------------------------------------------------------
static constexpr int RF_T = 1024, RF_W = RF_T / 64, RF_CAP = 4096, RF_SPLIT = 64;
static constexpr int S8_DEGENERATE = 12;           // candidates in one slice (1 / RF_SPLIT of the rows) beyond which the byte image is not worth scanning (~770 in all)

__global__ __launch_bounds__(256) void mocha_rowresid(const float* __restrict__ x, const float* __restrict__ centre,
                                                      const unsigned short* __restrict__ x16, float* __restrict__ rho, int cols) {
    __shared__ float red[2][4];
    const size_t row = blockIdx.x;
    float a = 0.f, nn = 0.f;
    for (int i = threadIdx.x; i < cols; i += 256) {
        const float t = x[row * cols + i] - centre[i];
        const float d = t - __uint_as_float((unsigned)x16[row * cols + i] << 16);
        a = fmaf(d, d, a); nn = fmaf(t, t, nn);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { a += __shfl_xor(a, o); nn += __shfl_xor(nn, o); }
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = a; red[1][threadIdx.x >> 6] = nn; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const float r2 = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]), n2 = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
        rho[row] = sqrtf(r2) * 1.000001f + 1e-6f * sqrtf(n2);
    }
}

hipError_t launch_rowresid(const float* x, const float* centre, const void* x16, float* rho, int64_t rows, int cols, hipStream_t s) {
    if (rows <= 0) return hipSuccess;
    hipLaunchKernelGGL(mocha_rowresid, dim3((unsigned)rows), dim3(256), 0, s, x, centre, (const unsigned short*)x16, rho, cols);
    return hipGetLastError();
}

__device__ __forceinline__ float key_value(unsigned long long k) {
    unsigned u = (unsigned)(k >> 32);
    u = (u & 0x80000000u) ? (u & 0x7fffffffu) : ~u;
    return __uint_as_float(u);
}

__global__ __launch_bounds__(RF_T) void mocha_match_refine(const unsigned long long* __restrict__ keys /*[Q][N]: coarse d16^2 keys*/,
                                                           const unsigned long long* __restrict__ wgmin /*[Q][nwg]: the scan's per-workgroup minima*/,
                                                           int nwg, const float* __restrict__ rho_bf16, const float* __restrict__ bank /*fp32 rows*/,
                                                           const float* __restrict__ query /*fp32, uncentred*/, long long N, int D,
                                                           int32_t* __restrict__ idx, float* __restrict__ dist,
                                                           unsigned long long* __restrict__ part /*[Q][RF_SPLIT]*/, unsigned* __restrict__ ticket /*[Q]*/,
                                                           const float* __restrict__ rho8 = nullptr, unsigned* __restrict__ mode = nullptr) {
    // match_scan8.hip: the keys came from the byte image (mode[0] == 0: their bounds are rho8) or from the bf16 copy; a slice with more than
    // S8_DEGENERATE candidates under the byte image's bounds switches the NEXT calls to the bf16 scan (mode[1], sticky until mocha_bank_set)
    const bool from8 = rho8 && mode && mode[0] == 0;
    const float* __restrict__ rho = from8 ? rho8 : rho_bf16;
    int slice_cands = 0;
    extern __shared__ __attribute__((aligned(16))) float rf_q[];       // [D]
    __shared__ unsigned long long rk[RF_W];
    __shared__ unsigned long long r_key;
    __shared__ int cand[RF_CAP];
    __shared__ int ncand;
    __shared__ float dsum[2][RF_W];
    __shared__ unsigned long long best;
    __shared__ int is_last;
    const int q = blockIdx.x, sl = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) { ncand = 0; best = ~0ull; }
    // the query row goes global -> LDS while the minima are read (LDS-DMA, 1 KB per wave instruction)
    const size_t qo = (size_t)q * D;
    for (int pi = wave; pi * 256 < D; pi += RF_W)
        __builtin_amdgcn_global_load_lds(query + qo + pi * 256 + lane * 4, (__attribute__((address_space(3))) void*)(rf_q + pi * 256), 16, 0, 0);
    // this workgroup's slice of the rows; up to RF_KPT keys and residual bounds per thread are fetched NOW, beside the minima
    const unsigned long long* kq = keys + (size_t)q * N;
    const long long per_slice = (N + RF_SPLIT - 1) / RF_SPLIT;
    const long long lo = (long long)sl * per_slice, hi_row = lo + per_slice < N ? lo + per_slice : N;
    constexpr int RF_KPT = 2;
    const bool in_regs = per_slice <= (long long)RF_KPT * RF_T;
    unsigned long long kreg[RF_KPT]; float rreg[RF_KPT];
#pragma unroll
    for (int u = 0; u < RF_KPT; ++u) {
        const long long n = lo + tid + (long long)u * RF_T;
        const bool ok = in_regs && n < hi_row;
        kreg[u] = ok ? kq[n] : ~0ull;
        rreg[u] = ok ? rho[n] : 0.f;
    }
    // ---- 1. the coarse minimum, from the scan's per-workgroup minima
    const unsigned long long* wq = wgmin + (size_t)q * nwg;
    unsigned long long key = ~0ull;
    for (int i = tid; i < nwg; i += RF_T) { const unsigned long long k = wq[i]; key = k < key ? k : key; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const unsigned long long k2 = __shfl_xor(key, o); key = k2 < key ? k2 : key; }
    if (lane == 0) rk[wave] = key;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                   // this wave's pieces of the query row have landed ...
    __syncthreads();                                                  // ... and everyone else's
    if (tid == 0) {
        unsigned long long k = rk[0];
        for (int w = 1; w < RF_W; ++w) k = rk[w] < k ? rk[w] : k;
        r_key = k;
    }
    __syncthreads();
    const unsigned nmin = (unsigned)(r_key & 0xffffffffull);
    const float d16min = sqrtf(fmaxf(key_value(r_key), 0.f));
    // (no finite coarse key at all - every key NaN or ~0 - leaves nmin out of range: nothing is excluded then)
    const float hi = (long long)nmin < N ? d16min + rho[nmin] + 2e-5f * d16min : __builtin_inff();      // upper bound of the coarse minimum's exact distance
    auto qualifies = [&](long long n, unsigned long long k, float r) -> bool {
        if ((unsigned)n == nmin) return true;
        const float d = sqrtf(fmaxf(key_value(k), 0.f));
        return d - r - 2e-5f * d <= hi;                                // NaN distances never qualify
    };
    // ---- 2. index windows of RF_CAP rows of the slice: list, then exact distances.  EVERY wave takes part in EVERY row: the row's
    // D / 4 pieces are cut into RF_W fixed segments, wave w sums segment w (lanes stride it, then a wave reduction) and the RF_W segment
    // sums are added in segment order - the result does not depend on how many rows are in the list, in this slice or in any other,
    // so identical rows get identical distances wherever they sit (ties then go to the lowest row).  Two rows are in flight per wave.
    const int seg = (D / 4) / RF_W;                                     // pieces per segment (D % 256 == 0: a whole number)
    const f32x4* qseg = reinterpret_cast<const f32x4*>(rf_q) + wave * seg;
    for (long long w0 = lo; w0 < hi_row; w0 += RF_CAP) {
        __syncthreads();
        if (tid == 0) ncand = 0;
        __syncthreads();
        if (in_regs) {
#pragma unroll
            for (int u = 0; u < RF_KPT; ++u) {
                const long long n = lo + tid + (long long)u * RF_T;
                if (n >= w0 && n < w0 + RF_CAP && n < hi_row && qualifies(n, kreg[u], rreg[u])) cand[atomicAdd(&ncand, 1)] = (int)n;
            }
        } else {
            for (long long n = w0 + tid; n < w0 + RF_CAP && n < hi_row; n += RF_T)
                if (qualifies(n, kq[n], rho[n])) cand[atomicAdd(&ncand, 1)] = (int)n;
        }
        __syncthreads();
        const int nc = ncand;
        slice_cands += nc;
        for (int p0 = 0; p0 < nc; p0 += 2) {
            const int nb = nc - p0 < 2 ? nc - p0 : 2;
            constexpr int NL = 6;                                       // 16-byte loads per lane and row: covers segments of up to 384 pieces
            f32x4 wv_[2][NL];
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const int row = cand[p0 + (c < nb ? c : 0)];
                const f32x4* b = reinterpret_cast<const f32x4*>(bank + (size_t)row * D) + wave * seg;
#pragma unroll
                for (int u = 0; u < NL; ++u) {
                    int pc = u * 64 + lane;
                    pc = pc < seg ? pc : seg - 1;
                    wv_[c][u] = __builtin_nontemporal_load(b + pc);
                }
            }
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                float a = 0.f;
#pragma unroll
                for (int u = 0; u < NL; ++u) {
                    const int pc = u * 64 + lane;
                    if (pc < seg) {
                        const f32x4 d = qseg[pc] - wv_[c][u];
                        a = fmaf(d[0], d[0], a); a = fmaf(d[1], d[1], a); a = fmaf(d[2], d[2], a); a = fmaf(d[3], d[3], a);
                    }
                }
                for (int pc0 = NL * 64; pc0 < seg; pc0 += 64) {         // longer rows (not the path's D): the rest of the segment
                    const int pc = pc0 + lane;
                    if (pc < seg && c < nb) {
                        const f32x4 d = qseg[pc] - reinterpret_cast<const f32x4*>(bank + (size_t)cand[p0 + c] * D)[wave * seg + pc];
                        a = fmaf(d[0], d[0], a); a = fmaf(d[1], d[1], a); a = fmaf(d[2], d[2], a); a = fmaf(d[3], d[3], a);
                    }
                }
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o);
                if (lane == 0) dsum[c][wave] = a;
            }
            __syncthreads();
            if (tid == 0) {
                unsigned long long bk = best;
                for (int c = 0; c < nb; ++c) {
                    float d2 = 0.f;
                    for (int sw = 0; sw < RF_W; ++sw) d2 += dsum[c][sw];
                    // distances are >= 0 (or NaN, which sorts last): the plain bit pattern orders them
                    const unsigned long long k = ((unsigned long long)__float_as_uint(d2) << 32) | (unsigned)cand[p0 + c];
                    bk = (bk == ~0ull) || k < bk ? k : bk;
                }
                best = bk;
            }
            __syncthreads();
        }
    }
    // ---- 3. the slices' winners: each workgroup publishes its own, the one that arrives last picks the smallest (distance, row) key.
    // The ticket counts this launch's arrivals; every launch runs all RF_SPLIT workgroups of every query, so the last arriver sees
    // RF_SPLIT - 1 and puts the word back to zero for the next launch (it never wraps).  This is NOT a recovery mechanism: a launch that did
    // not complete (a device fault) leaves the ticket at some k != 0 and the scratch head must then be zeroed again by the host
    // (ensure_match_scratch does so whenever it allocates the buffer).  Calls that share a scratch buffer are ordered by their stream.
    __syncthreads();
    if (tid == 0 && from8 && slice_cands > S8_DEGENERATE) atomicExch(mode + 1, 1u);
    if (tid == 0) {
        __atomic_store_n(part + (size_t)q * RF_SPLIT + sl, best, __ATOMIC_RELAXED);
        __threadfence();
        const unsigned t = atomicAdd(ticket + q, 1u);
        is_last = (t % RF_SPLIT) == RF_SPLIT - 1;
    }
    __syncthreads();
    if (is_last && tid == 0) {
        __threadfence();
        unsigned long long bk = ~0ull;
        for (int i = 0; i < RF_SPLIT; ++i) {
            const unsigned long long k = __atomic_load_n(part + (size_t)q * RF_SPLIT + i, __ATOMIC_RELAXED);
            bk = k < bk ? k : bk;
        }
        idx[q] = (int)(bk & 0xffffffffull);
        if (dist) dist[q] = sqrtf(__uint_as_float((unsigned)(bk >> 32)));
        atomicExch(ticket + q, 0u);
    }
}

// words of scratch launch_match_scan16 needs for a bank of N rows: the slices' winners and the arrival tickets (the head: must be
// ZERO when the buffer is first used), then 8 queries' keys
size_t match_scan16_scratch_head_words() { return 8 * RF_SPLIT + 8 + 1; }      // winners, tickets, the byte stage's two mode words
size_t match_scan8_mode_word() { return 8 * RF_SPLIT + 8; }
static size_t scan_nwg(int64_t N) { const int r = MS_WAVES * MS_ROWS_PER_WAVE; return (size_t)((N + r - 1) / r); }
size_t match_scan16_scratch_words(int64_t N) { return match_scan16_scratch_head_words() + (size_t)8 * N + (size_t)8 * scan_nwg(N); }

hipError_t match_refine_init() {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(&mocha_match_refine), hipFuncAttributeMaxDynamicSharedMemorySize, 23040 * 4);
}

// the refine launch alone (match_scan8.hip scans with its own kernel): keys / wgmin as the scan left them, nq <= 8
hipError_t launch_match_refine(const unsigned long long* keys, const unsigned long long* wgmin, int nwg, const float* rho16, const float* rho8, unsigned* mode,
                               const float* bank, const float* query, int nq, int64_t N, int D, int32_t* idx, float* dist, unsigned long long* scratch, hipStream_t s) {
    unsigned long long* part = scratch;
    hipLaunchKernelGGL(mocha_match_refine, dim3(nq, RF_SPLIT), dim3(RF_T), (size_t)D * sizeof(float), s, keys, wgmin, nwg, rho16, bank, query,
                       (long long)N, D, idx, dist, part, reinterpret_cast<unsigned*>(part + 8 * RF_SPLIT), rho8, mode);
    return hipGetLastError();
}

// few queries against an fp32 bank through its centred bf16 copy: `qc` = the queries minus the centroid (scan operand), `query` the
// queries themselves (exact re-rank); scratch: match_scan16_scratch_words(N) words, head zeroed once
hipError_t launch_match_scan16(const void* bank16, const float* rho, const float* bank, const float* qc, const float* query, int Q, int64_t N,
                               int D, unsigned long long* scratch, int32_t* idx, float* dist, hipStream_t s) {
    if (Q <= 0) return hipSuccess;
    // scratch: [slice winners 8 x RF_SPLIT][arrival tickets 8][coarse keys 8 x N][the scan workgroups' minima 8 x N / 16] - the head at a fixed place whatever N
    unsigned long long* part = scratch;
    unsigned long long* keys = scratch + match_scan16_scratch_head_words();
    unsigned long long* wgmin = keys + (size_t)8 * N;          // [8][workgroups of the scan]
    if (D % MS_CHUNK_BF16 != 0 || D % 256 || D > 23040 || N < 1 || N > 0x7ffffff0ll) return hipErrorInvalidValue;
    const int rows_per_wg = MS_WAVES * MS_ROWS_PER_WAVE;
    const unsigned grid = (unsigned)((N + rows_per_wg - 1) / rows_per_wg);
    for (int q0 = 0; q0 < Q; q0 += 8) {
        const int nq = (Q - q0) < 8 ? (Q - q0) : 8;
        const float* qp = qc + (size_t)q0 * D;
#define MS16_LAUNCH(QQ) hipLaunchKernelGGL((mocha_match_stream<QQ, true>), dim3(grid), dim3(256), 0, s, bank16, qp, nq, (long long)N, D, keys, 1, wgmin)
        if (nq == 1) MS16_LAUNCH(1);
        else if (nq == 2) MS16_LAUNCH(2);
        else if (nq <= 4) MS16_LAUNCH(4);
        else MS16_LAUNCH(8);
#undef MS16_LAUNCH
        hipLaunchKernelGGL(mocha_match_refine, dim3(nq, RF_SPLIT), dim3(RF_T), (size_t)D * sizeof(float), s, keys, wgmin, (int)grid, rho, bank, query + (size_t)q0 * D,
                           (long long)N, D, idx + q0, dist ? dist + q0 : nullptr, part, reinterpret_cast<unsigned*>(part + 8 * RF_SPLIT));
    }
    return hipGetLastError();
}

// fp32 -> bf16 (round to nearest even) copy of the matching bank, and its squared row norms are then
// taken from the rounded values (launch_rownorm2_bf16) so that value and norm stay consistent
__global__ __launch_bounds__(256) void mocha_to_bf16(const float* __restrict__ x, const float* __restrict__ sub, int cols4,
                                                     unsigned short* __restrict__ y, long long n4) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    f32x4 v = reinterpret_cast<const f32x4*>(x)[i];
    if (sub) v -= reinterpret_cast<const f32x4*>(sub)[i % cols4];          // centred bank: the rounding acts on b - centre
    unsigned short o[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const unsigned u = __float_as_uint(v[e]);
        o[e] = (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);      // inputs are finite features
    }
    reinterpret_cast<uint2*>(y)[i] = make_uint2((unsigned)o[0] | ((unsigned)o[1] << 16), (unsigned)o[2] | ((unsigned)o[3] << 16));
}

__global__ __launch_bounds__(256) void mocha_rownorm2_bf16(const unsigned short* __restrict__ x, float* __restrict__ out, int cols) {
    __shared__ float red[4];
    const size_t row = blockIdx.x;
    float a = 0.f;
    for (int i = threadIdx.x; i < cols; i += 256) {
        const float v = __uint_as_float((unsigned)x[row * cols + i] << 16);
        a = fmaf(v, v, a);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = a;
    __syncthreads();
    if (threadIdx.x == 0) out[row] = (red[0] + red[1]) + (red[2] + red[3]);
}

hipError_t launch_to_bf16(const float* x, const float* sub, int cols, void* y, int64_t n, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    if (n % 4 || cols % 4) return hipErrorInvalidValue;
    const long long n4 = n / 4;
    hipLaunchKernelGGL(mocha_to_bf16, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, x, sub, cols / 4, (unsigned short*)y, n4);
    return hipGetLastError();
}

hipError_t launch_rownorm2_bf16(const void* x, float* out, int64_t rows, int cols, hipStream_t s) {
    if (rows <= 0) return hipSuccess;
    hipLaunchKernelGGL(mocha_rownorm2_bf16, dim3((unsigned)rows), dim3(256), 0, s, (const unsigned short*)x, out, cols);
    return hipGetLastError();
}

}  // namespace mocha
