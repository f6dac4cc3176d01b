// Internal launcher interface between the host-side context (mocha_api.cpp) and the gfx950
// kernels (*.hip).  Not part of the C ABI (include/mocha_hip.h is).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mocha {

// ---------------------------------------------------------------------------------------
// fp32 MFMA GEMM:  C[M,N] = epilogue( Aop[M,K] · W[N,K]^T )
//
// Aop is either the plain row-major matrix A (lda) or an on-the-fly gather of a
// (batch, time, node, channel) activation tensor that realises a temporal convolution with
// reflect padding as a GEMM (net/blocks.py:112-118): for output row (b, t, v) and
// k = tap*Cc + c the operand is
//      ascale * sum_{j<R} src[(b, refl(t*stride + j + tap*tstep - pad, T_full) >> tshift, v), c]
// R = 4, stride = 4 additionally folds AvgPool2d((4,1)) (model.py:47) into the operand;
// tshift = 2 reads a nearest-x4-upsampled tensor (model.py:74) without materialising it; tstep = 4 with stride = 4 walks that tensor's
// SOURCE frames (taps a source frame apart; the reflection at the upsampled ends is then a clamp on source frames: mocha_api.cpp, fold_upsample).
// ---------------------------------------------------------------------------------------
struct GemmParams {
    const float* A = nullptr;     // source activations
    const float* W = nullptr;     // [N][K], k contiguous (nn.Linear / repacked conv layout)
    const float* wsub = nullptr;              // [K] vector subtracted from every W row while it is staged (centred bank)
    const unsigned short* Wsplit = nullptr;   // gemm_x3.hip: the packed three-plane image of W (launch_pack_x3)
    float* C = nullptr;
    const float* bias = nullptr;      // [N] or null
    const float* rowbias = nullptr;   // [rb_mod][N] or null, indexed by (row % rb_mod)
    const float* residual = nullptr;  // [M][ldr] or null, added after the activation
    int M = 0, N = 0, K = 0;
    int lda = 0, ldc = 0, ldr = 0;
    int rb_mod = 1;
    int act = 0;          // 0 none, 1 exact-erf GELU, 2 LeakyReLU(0.2), 3 ReLU
    int a_lrelu = 0;      // LeakyReLU(0.2) applied to the A operand as it is loaded
    int gather = 0;       // 0 plain rows, 1 temporal gather
    int T_out = 1, V = 1, ntaps = 1, pad = 0, stride = 1, R = 1, T_full = 1, tshift = 0, Cc = 0, T_src = 1;
    int tstep = 1;        // distance between taps on the T_full time line
    float ascale = 1.f;
    int ksplit = 1;               // >1: split K over gridDim.z, raw partial sums to C + z*slab_stride
    long long slab_stride = 0;
    // gemm_x3.hip, per context (options "gemm_persistent", "gemm_persistent_max_n"): workgroups of the persistent instance (0 = never; a
    // multiple of 8) and the widest launch, in columns, that takes it
    int persistent = 768, persistent_max_n = 512;
    int tile64_below = 0;         // gemm_x3.hip: mid-size launches with fewer 64 x 128 tiles than this take 64 x 64 tiles (option "gemm_tile64_below")
    // gemm_h2.hip (two fp16 planes, three passes; option "gemm_f16x2"): the packed two-plane image of W and 1 / S_w per row (launch_pack_h2);
    // rows_per_win: consecutive rows of the launch that form one WINDOW (the path's independent unit; >= 64); a_amax: per window a bound on
    // |A| (after a_lrelu), a vector in DEVICE memory - the window's activation scale comes from it; c_amax: null, or the vector where the
    // epilogue leaves, per window, the largest magnitude it stores (atomic max; zeroed by the caller) - a later launch's a_amax
    const unsigned short* Wh2 = nullptr;
    const float* w_inv = nullptr;
    const float* a_amax = nullptr;
    float* c_amax = nullptr;
    int rows_per_win = 0;
};
hipError_t launch_gemm(const GemmParams& p, hipStream_t s);
bool gemm_is_narrow(const GemmParams& p);
bool gemm_is_small(const GemmParams& p);    // true: mid-size launch -> 64 x 64 tiles
bool gemm_is_skinny(const GemmParams& p);
bool gemm_is_skinny16(const GemmParams& p); // true: one or two windows -> mocha_gemm_skinny16 (16x16 tile per workgroup, every load of a wave's K quarter in flight at once)   // true: a handful of windows -> mocha_gemm_skinny (32x32 tile per workgroup, 4-way in-workgroup split-K)
hipError_t gemm_init();           // one-time function attributes (dynamic LDS size)
// fp32 GEMM on the bf16 matrix pipe (gemm_x3.hip): both operands as three bf16 planes, six MFMA passes, fp32-accurate.
// p.Wsplit = the packed image of W made by launch_pack_x3 (gemm_x3_packed_elems(N, K) bf16).
hipError_t gemm_x3_init();
bool gemm_x3_supports(const GemmParams& p);
size_t gemm_x3_packed_elems(int N, int K);
hipError_t launch_pack_x3(const float* W, int N, int K, unsigned short* out, hipStream_t s, const float* wsub = nullptr);   // wsub: K values subtracted from every row
hipError_t launch_gemm_x3(const GemmParams& p, hipStream_t s);
// the same engine with the activations resident in registers (gemm_x3r.hip, round 6): K = 256, N a multiple of 128 (>= 256), plain rows, bias / activation epilogue;
// bit-identical to launch_gemm_x3.  grid: workgroups (one per CU; 0 = 256)
hipError_t gemm_x3r_init();
bool gemm_x3r_supports(const GemmParams& p);
hipError_t launch_gemm_x3r(const GemmParams& p, hipStream_t s, int grid = 0);
// fp32 GEMM on the fp16 matrix pipe (gemm_h2.hip): both operands as two fp16 planes with power-of-two scales, three MFMA passes.
hipError_t gemm_h2_init();
bool gemm_h2_supports(const GemmParams& p);
size_t gemm_h2_packed_elems(int N, int K);
hipError_t launch_pack_h2(const float* W, int N, int K, unsigned short* out, float* w_inv /*N*/, hipStream_t s);
hipError_t launch_gemm_h2(const GemmParams& p, hipStream_t s);      // needs p.Wh2, p.w_inv, p.a_amax
hipError_t launch_absmax(const float* x, long long nwin, long long per, float* out /*nwin, zeroed by the caller*/, hipStream_t s, float mul = 1.f, float add = 0.f);   // out[w] = max(out[w], mul * max|window w of x| + add)
hipError_t launch_gather_f32(const float* table, const int32_t* idx, long long rows, float* out, int n, hipStream_t s);    // out[i] = table[clamp(idx[i])]


// ---------------------------------------------------------------------------------------
// Multi-head attention, nq queries x nk keys (<= 192), one workgroup per (window, head)
// (net/transformer.py:65-76; nn.MultiheadAttention in model_CVAE.py):
//   out[b, i, h*DH + d] = softmax_j(q_i·k_j * scale) v_j ;  batch b starts at row b*nq (q, out) / b*nk (k, v)
// ---------------------------------------------------------------------------------------
struct AttnParams {
    const float* q; const float* k; const float* v; float* out;
    int ldq, ldk, ldv, ldo;       // row strides in floats
    int B, heads, dh, nq, nk;
    float scale;
    int hsk = -1, hsv = -1;       // column offset per head of k / v (default dh; 0 = all heads share the same rows)
    int split_max = 192;          // (window, head) pairs up to which the twelve-wave head-dim-256 variant is launched (0 = never)
    // launch_attention_x3 only: window b's keys / values are rows [kv_idx[b] * nk, +nk) of k / v (index clamped to [0, kv_rows)) - the decoder
    // reading IN(cha) and cha of the matched bank entries in place (test_fullframework.py:465), no gathered copy
    const int32_t* kv_idx = nullptr; long long kv_rows = 0;
};
hipError_t launch_attention(const AttnParams& p, hipStream_t s);
// the same on the bf16 matrix pipe with exact three-plane operands (attention_x3.hip): nq, nk <= 96, dh = 128 / 256
hipError_t launch_attention_x3(const AttnParams& p, hipStream_t s);
hipError_t attention_x3_init();              // one-time function attributes (dynamic LDS of the twelve-wave small-batch variant)
// Cross-attention whose keys and values are the same for every head (the folded decoder: K = IN(cha), V = cha, dh = 256), from the
// pre-split bf16 plane images mocha_instnorm writes (InormExtra::kvimg; layout: attention_kv.hip).  One workgroup per (window, head pair).
static constexpr int ATTN_KV_STAGE_BYTES = 3 * 96 * 64;               // three planes x 96 rows x 32 bf16
static constexpr int ATTN_KV_IMG_BYTES = 16 * ATTN_KV_STAGE_BYTES;    // 8 K stages + 8 V stages per window: 294 912 B
struct AttnKvParams {
    const float* q; float* out; const unsigned short* kv;
    int ldq, ldo;                 // row strides in floats
    int B, heads, dh, nq, nk;
    float scale;
    int pairs = 0;                // 1: a six-wave workgroup per head pair even when the heads come in fours (diagnostic)
};
hipError_t launch_attention_x3_kv(const AttnKvParams& p, hipStream_t s);

// ---------------------------------------------------------------------------------------
// Small bandwidth-bound kernels
// ---------------------------------------------------------------------------------------
// X (B,T,V,Cin) -> 1x1 conv Cin->64 + bias -> LeakyReLU -> hop-partitioned adjacency -> joint->part pool,
// written as rows (b,t,p) x (k*64+c)   [model.py:44-46 front half, net/blocks.py:57-66,131]
// raw_root = 1: X frames are (V+1, Cin) with the root bone first and are z-scored with xmean/xstd ((V+1)*Cin) on load
hipError_t launch_embed_front(const float* X, const float* W1, const float* b1, const float* AP /*3*V*6*/,
                              float* out, int nframes, int V, int Cin, const float* xmean, const float* xstd, int raw_root,
                              hipStream_t s, bool planes = false, int max_wgs = 512);
hipError_t launch_embed_sums(const float* X, const float* W1, const float* b1, const float* AP, float* u, int nwin, int V, int Cin,
                             const float* xmean, const float* xstd, int raw_root, hipStream_t s, int max_wgs = 512);      // max_wgs: option "embed_front_max_wgs" (per context)
// rows (b,t,p) x 256 -> LeakyReLU -> body-part adjacency (2 hops) -> rows (b,t,w) x (k*256+c)
hipError_t launch_body_front(const float* x, const float* A_b /*2*6*6*/, float* out, int rows6 /*B*15*/, hipStream_t s);
// g rows (b,t',p) x (k*64+c) -> y2c rows (b,t',w) x 64 : sum_k sum_p AU[k][p][w] g[...]
hipError_t launch_joint_expand(const float* g, const float* AU /*3*6*V*/, float* out, int nframes15, int V, hipStream_t s);
// z rows x 64 -> LeakyReLU -> 1x1 conv 64->Cout + bias -> Y rows x Cout   [model.py:77-79]
// ymean/ystd ((V+1)*Cout, root row first) non-null: Y is de-normalised in the epilogue
hipError_t launch_final_proj(const float* z, const float* W6, const float* b6, float* Y, int rows, int Cout, int V,
                             const float* ymean, const float* ystd, hipStream_t s, int phased = 0);
// per (b, channel) instance norm over n tokens (net/transformer.py:13-20).
//   out = (x-mean)/(std+eps); mean_out (B,256) optional; zn = (out - gm)/gs optional
// optional extras of the instance norm: zc = zn - centre (the matcher's centred queries, bit-identical to mocha_sub_rows on zn);
// rows gathered from a table (x row of window b = table[clamp(row_idx[b])], the decoder's cha_encoded[frame_index]) and copied out
static constexpr int QSTAT_PARTS = 4;      // row statistics come in parts (one per workgroup of a window's channel quarter), added in order by mocha_match_select2
struct InormExtra {
    const float* centre = nullptr; float* zc = nullptr;      // zc: (z-score - centre), fp32
    unsigned short* zc16 = nullptr;                          // the same as bf16 (round to nearest even): the many-query bf16 pass's query plane
    // zc16 with plane_stride > 0: a SECOND plane bf16(zc - plane 0) at zc16 + plane_stride (elements) - the coarse pass then carries 16
    // significant bits of every query value; qstat (2 x QSTAT_PARTS floats per window, summed by the consumer in part order): ||zc||^2 and the squared norm of what the planes (zc16) leave out
    // - or, with zc alone, {||zc||^2, 0} - the selection's error bound (match_select2.hip) without a pass over the row
    long long plane_stride = 0; float* qstat = nullptr;
    const float* table = nullptr; const int32_t* row_idx = nullptr; long long table_rows = 0; float* copy_out = nullptr;
    // kvimg: per window the pre-split key / value images of launch_attention_x3_kv (ATTN_KV_IMG_BYTES each): K = the normalised rows,
    // V = the input rows, three bf16 planes each, rows n .. 95 zero
    unsigned short* kvimg = nullptr;
    int split_max = 1 << 30;             // windows up to which a window's channels go over four workgroups (per context: option "inorm_split_max")
    double* mean64 = nullptr;            // (B, 256) token mean summed in float64: the input of the float64 style MLP (launch_linear_f64)
};
hipError_t launch_instnorm(const float* x, float* out, float* mean_out, const float* gm, const float* gs, float* zn,
                           int B, int n, hipStream_t s, const InormExtra* ex = nullptr);
// AdaIN + the attention's mapping norm (net/transformer.py:108-113, 49-56):
//   xad = (1+gamma)*IN(x)+beta ; qin = IN(xad) ; gb (B,512) = [gamma | beta]
// closed != 0: qin from the first statistics in closed form (no cancellation against beta; pointwise.hip)
// gb_idx non-null: window b reads its gamma / beta at gb + clamp(gb_idx[b], gb_rows) * gb_stride (the bank's cached style constants)
hipError_t launch_adain(const float* x, const float* gb, int gb_stride /*floats between windows*/, float* xad, float* qin, int B, int n, hipStream_t s,
                        int closed = 1, const int32_t* gb_idx = nullptr, long long gb_rows = 0, int split_max = 1 << 30);
// Y = act(X W^T + bias) in float64, L independent column blocks (pointwise.hip: the decoder's style MLP)
hipError_t launch_linear_f64(const double* X, int ldx, int xcol, const double* W, const double* bias, double* y64, float* y32, int ldy,
                             int M, int N, int K, int L, int act, hipStream_t s);
// u rows (b,t',p) x (dt*256+c) = 1/4 sum of the 4 reflect-indexed frames of tap dt (conv k=5 fused with AvgPool(4))
hipError_t launch_window_sums(const float* y, float* u, int rows, int channels /*256 or 192*/, hipStream_t s);
// ---- CVAE sampler pieces (cvae.hip; model_CVAE.py)
// y = LayerNorm(x) over 256 channels per row (eps 1e-5, biased variance), affine
hipError_t launch_layernorm256(const float* x, const float* w, const float* b, float* y, int rows, hipStream_t s);
// tokens (B,182,256) = [mu_token, logvar_token, c (B,180,256)] + pe[:182]          (model_CVAE.py:69-76)
hipError_t launch_cvae_prior_tokens(const float* c, const float* mu_tok, const float* lv_tok, const float* pe, float* out, int B, int nc, hipStream_t s);
// z = x[:,0] (+ eps * exp(0.5 x[:,1]));  mem (B,1+nc,256) = [z, c];  q (B,nq,256) = pe[:nq]   (model_CVAE.py:81-87,158-163)
hipError_t launch_cvae_latent(const float* x /*B,ntok,256*/, int ntok, const float* eps, const float* c, int nc, const float* pe, int nq,
                              float* mem, float* q, float* mu_out, float* logvar_out, int B, hipStream_t s);
// cond (B,2n,256) = cat[(src_cnt - sm)/ss, (prev - cm)/cs] over tokens; out = x*sd + mean   (test_fullframework.py:446-449)
hipError_t launch_cvae_condition(const float* src_cnt, const float* sm, const float* ss, const float* prev, const float* cm,
                                 const float* cs, float* cond, int B, int n, hipStream_t s);
hipError_t launch_scale_shift(const float* x, const float* mean, const float* sd, float* out, int B, int n, hipStream_t s);
// demo featurisation (featurize.hip): local bone features of B windows -> X (B,T,J,15) un-normalised, root bone included
hipError_t featurize_init();
hipError_t launch_featurize(const float* Yrot, const float* Ypos, const float* Yvel, const float* Yang, const int* parents /*J, device*/,
                            float* X, int B, int T, int J, hipStream_t s);
// post-processing of decoded windows (postprocess.hip)
#define MOCHA_MAX_CONTACT 4
#define MOCHA_MAX_CHAIN 8
#define MOCHA_MAX_BONES 32
struct PostParams {
    const float* heads;             // (clips, frames, V, 13)
    const float* speed;             // (clips, frames)
    const float* src_rvel;          // (clips, frames, 3)
    const float* src_rang;          // (clips, frames, 3)
    const float* src_speed;         // (clips, frames)
    const unsigned char* contact;   // (clips, frames, n_contact)
    double *pos, *rot, *ik_rot;     // (clips, frames, V+1, 3|4|4)
    double *bvh_pos, *bvh_euler;    // (clips, frames, V, 3) or null
    int n_clips, n_frames, V, n_contact, ik_enabled, blend_enabled;
    int parents[MOCHA_MAX_BONES];
    int contact_bones[MOCHA_MAX_CONTACT];
    double dt, max_length_buffer, foot_height, unlock_radius, halflife;
};
hipError_t launch_pose_heads(const float* Y, float* heads, float* speed, int B, int T, int V, hipStream_t s);
hipError_t launch_post_clip(const PostParams& p, hipStream_t s);
// per-column mean and population std over N rows (bank build: cnt_norm)
size_t column_mean_scratch_doubles(int cols);
hipError_t launch_column_mean(const float* x, int64_t N, int cols, float* mean, double* scratch, hipStream_t s);
hipError_t launch_column_stats(const float* x, int64_t N, int cols, float* mean, float* sd, hipStream_t s);
// bank row squared norms
hipError_t launch_rownorm2(const float* x, const float* sub /*or null*/, float* out, int64_t rows, int cols, hipStream_t s);
hipError_t launch_sub_rows(const float* x, const float* sub, float* out, int64_t rows, int cols, hipStream_t s);
// many-query matcher (match_mfma.hip): centred bf16 queries, one-plane bf16 coarse scores S[z][q][n] (K split z), and the
// select kernel: per query, EVERY row whose coarse score ||b-c||^2 - 2 S lies within margin_rel * (2||q-c||^2 + ||b-c||^2 +
// ||b0-c||^2) of the best one (margin_rel >= 0 bounds the coarse pass's error; a negative value is rejected) is re-evaluated
// exactly in the direct form; the smallest exact distance wins, ties to the lowest row.
// bank = raw fp32 rows (exact distance sum (q - b)^2) or bank16 = centred bf16 rows (sum ((q - c) - b16)^2).
hipError_t match_mfma_init();
int match_bf16_ksplit(int Q, int64_t N);
hipError_t launch_center_bf16(const float* x, const float* centre, void* out, int64_t rows, int cols, hipStream_t s);
// row by row: planes[0] = bf16(x - centre) and, with nplanes = 2, planes[1] = bf16(x - centre - planes[0]) (stacked: plane 1 starts rows * cols
// elements after plane 0), or out32 = x - centre in fp32; qstat[row] = QSTAT_PARTS pairs whose sums are {||x - centre||^2, ||x - centre - planes||^2 (0 for fp32)} (all of it in part 0).  One
// workgroup per row, fixed summation order.  Exactly one of planes / out32 is non-null.
hipError_t launch_center_rows(const float* x, const float* centre, void* planes, int nplanes, float* out32, float* qstat, int64_t rows, int cols, hipStream_t s);
// planes = 2: qc16 holds two stacked bf16 planes of the queries (plane 1 starts Q * D elements after plane 0), S = (a0 + a1) b^T
// tiled: the bank as launch_tile_bf16 wrote it (match_tiled_elems(N, D) bf16), or null to read the row-major bank16
// nt_bank != 0: the bank's LDS-DMA loads carry the non-temporal hint
hipError_t launch_match_gemm_bf16(const void* qc16, const void* bank16, float* S, int Q, int64_t N, int D, int ksplit, hipStream_t s, int planes = 1,
                                  const void* tiled = nullptr, int nt_bank = 0, unsigned* tickets = nullptr);
// tickets (option "match_fold", round 6): ceil(Q / 128) * ceil(N / 128) zeroed words; the last-arriving K-slab workgroup of a tile leaves the slabs' sum in slab 0
// round 5 (match_pass.hip): 128 queries x 256 rows per workgroup, the bank straight into registers, only the queries through LDS;
// variant: bits 0-3 register prefetch depth in 64-k stages (0 = default), bit 4 non-temporal bank loads, bit 8 fill only (measurement)
int match_pass256_ksplit(int Q, int64_t N);
hipError_t launch_match_pass256(const void* qc16, const void* bank16, float* S, int Q, int64_t N, int D, int ksplit, hipStream_t s, int variant = 0,
                                int planes = 1, const void* tiled32 = nullptr);
// the bank in the pass's operand order (every wave-level load 1 KB contiguous): match_tile32_elems(N, D) bf16
size_t match_tile32_elems(int64_t N, int D);
hipError_t launch_tile32_bf16(const void* bank16, void* out, int64_t N, int D, hipStream_t s);
size_t match_tiled_elems(int64_t N, int D);
hipError_t launch_tile_bf16(const void* bank16, void* out, int64_t N, int D, hipStream_t s);
hipError_t launch_match_select(const float* S, int ksplit, long long slab_stride, int lds, const float* bnorm, const float* query,
                               const float* centre, const float* bank, const void* bank16, float margin_rel, int Q, int64_t N, int D,
                               int32_t* idx, float* dist, hipStream_t s);
// the same selection without the usually unnecessary work (match_select2.hip): the row statistics of the error bound come from the
// producer of the centred queries (qstat: 2 x QSTAT_PARTS floats per query - launch_center_rows, InormExtra::qstat), nothing is staged in LDS, and a
// query whose coarse minimum stands alone is answered without touching a bank row when dist == nullptr.  Otherwise the same arguments,
// bounds and result semantics as launch_match_select.
hipError_t launch_match_select2(const float* S, int ksplit, long long slab_stride, int lds, const float* bnorm, const float* query,
                                const float* centre, const float* bank, const void* bank16, const float* qstat, float margin_rel, int Q,
                                int64_t N, int D, int32_t* idx, float* dist, hipStream_t s);
// streaming matcher for few queries against a large bank (HBM-bound): bank fp32 or bf16;
// partial = match_stream_scratch(Q, N) u64 words of scratch
size_t match_stream_scratch(int Q, int64_t N);
hipError_t launch_match_stream(const void* bank, int bank_bf16, const float* query, int Q, int64_t N, int D,
                               unsigned long long* partial, int32_t* idx, float* dist, hipStream_t s);
// k nearest rows per query, exact (every row's direct-form distance, then a k-pass selection); keys = 8 * N u64 words of scratch
hipError_t launch_match_topk(const void* bank, int bank_bf16, const float* query, int Q, int64_t N, int D, int k,
                             unsigned long long* keys, int32_t* idx, float* dist, hipStream_t s);
// out[q] = sum_j softmax_j(-dist[q][j] / temperature) * src[idx[q][j]] over the k neighbours of query q
hipError_t launch_gather_blend(const float* src, const int32_t* idx, const float* dist, float temperature, float* out, int Q, int k,
                               int cols, int64_t nrows, hipStream_t s);
hipError_t launch_to_bf16(const float* x, const float* sub /*per-column, or null*/, int cols, void* y, int64_t n, hipStream_t s);
// fp32 bank scanned through its centred bf16 copy, exact re-rank of what the rounding cannot exclude (match_stream.hip)
hipError_t match_refine_init();
size_t match_scan16_scratch_words(int64_t N);        // launch_match_scan16's scratch buffer, in 8-byte words; the first ..._head_words() must be zero at first use
size_t match_scan16_scratch_head_words();
// the one-byte first stage of the few-query scan (match_scan8.hip; option "scan8"): image + per-row scale + residual bound, the adaptive scan + refine
hipError_t launch_to_i8(const float* x, const float* centre, void* y, float* scale, float* rho, int64_t rows, int cols, hipStream_t s);
hipError_t match_scan8_init();
size_t match_scan8_mode_word();                      // word of the scan16 scratch head that holds the stage's two mode words {this call, next calls}
hipError_t launch_match_scan8(const void* bank8, const float* scale8, const float* rho8, const void* bank16, const float* rho16, const float* bank,
                              const float* qc, const float* query, int Q, int64_t N, int D, unsigned long long* scratch, int32_t* idx, float* dist, hipStream_t s);
hipError_t launch_match_refine(const unsigned long long* keys, const unsigned long long* wgmin, int nwg, const float* rho16, const float* rho8, unsigned* mode,
                               const float* bank, const float* query, int nq, int64_t N, int D, int32_t* idx, float* dist, unsigned long long* scratch, hipStream_t s);
hipError_t launch_rowresid(const float* x, const float* centre, const void* x16, float* rho, int64_t rows, int cols, hipStream_t s);
hipError_t launch_match_scan16(const void* bank16, const float* rho, const float* bank, const float* qc, const float* query, int Q, int64_t N,
                               int D, unsigned long long* scratch, int32_t* idx, float* dist, hipStream_t s);
hipError_t launch_rownorm2_bf16(const void* x, float* out, int64_t rows, int cols, hipStream_t s);
// out[q] = src[idx[q]] rows of `cols` floats
hipError_t launch_gather_rows(const float* src, const int32_t* idx, float* out, int Q, int cols, int64_t nrows, hipStream_t s);

}  // namespace mocha
