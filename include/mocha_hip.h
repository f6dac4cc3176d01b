/*
 * mocha_hip.h — C ABI of libmocha_hip.so: the MI355X (gfx950) implementation of the MOCHA
 * Generator hot path (motion encoder -> context matching -> body-part decoder).
 *
 * The reference (DK-Jang/MOCHA_SIGASIA2023) is pure Python and has no FFI of its own; the
 * boundary it offers is the attribute surface of its `Generator` module plus the
 * `gen_ema` state_dict key schema.  Each entry point below names the
 * reference interface it replaces (file:line relative to the reference repository root).
 * INTEGRATION.md shows the ctypes binding a maintainer of the reference would add.
 *
 * Conventions
 *   - every tensor pointer is a DEVICE pointer to contiguous fp32 unless a comment says host;
 *   - layouts are the reference's: poses (B, T, V, C_in) channel-last, tokens (B, 90, 256)
 *     with token = t*6 + body_part (model.py:49), bank entries likewise;
 *   - `stream` is a hipStream_t passed as void*; all work is enqueued on it.  Steady-state calls do not synchronise
 *     the device and are graph-capture safe; device allocations (and the device synchronisation that replacing a
 *     buffer needs) only happen in mocha_finalize_weights, mocha_reserve, mocha_bank_set (both also size the match
 *     scratch for every query count the workspace admits), mocha_bank_broadcast, mocha_set_option and the first call
 *     with a batch larger than any before.  Each such replacement bumps mocha_generation(ctx): a caller that captured
 *     calls into its own HIP graph compares the generation before replaying (mocha_step_graph does so itself);
 *   - a NULL where a required device pointer belongs is MOCHA_ERR_ARG ("... null argument"), never a launch; with an empty batch
 *     (B == 0) the tensor pointers may be NULL and nothing is touched (tests/test_edge_cases.py);
 *   - functions return 0 on success, a negative mocha_status otherwise;
 *     mocha_last_error(ctx) gives the message.  The caller owns every in/out buffer; the
 *     context owns the device copies of the weights, the bank (unless borrowed) and the
 *     workspaces.  One context per device per host thread.
 *   - several contexts of one process may be driven concurrently from different streams (tests/test_runtime_gpu.py runs two whole
 *     pipelines side by side on same-priority and on different-priority streams and compares them bit for bit with the sequential
 *     results); a context itself serves one call at a time.
 *   - a platform hazard found in round 2, for applications that run THEIR OWN kernels on another stream while this library works:
 *     a kernel that issues ordinary VALU instructions between bf16 MFMAs (the plane engines do) was observed to corrupt the results
 *     of `v_pk_fma_f32 ... op_sel` (low result from a high VGPR) in OTHER kernels sharing its CU - lanes 48-63, value 0
 *     (tools/body_front_repro.hip reproduces it without this library's pipeline).  What is established by test, not by guess
 *     (tests/test_concurrency_stress.py, run on every GPU test pass): under that aggressor - large-batch plane GEMMs and plane
 *     attention streaming from a second context - the round-2 build of the affected kernel (kept as a test-only canary) differs
 *     from its solo result in more than half of 3 000 repetitions, while EVERY kernel class of this library, each repeated 1 000
 *     times on same- and different-priority streams, reproduces its solo result bit for bit - including the kernels that use
 *     packed fp32 arithmetic by design (mocha_match_stream) or by the compiler's choice (GEMM epilogues, window sums, norms).
 *     tools/isa_lint.py additionally keeps the one pattern that failed (low result from the HIGH register of a VGPR pair) out
 *     of the library's device code.  If foreign kernels must overlap with the library's calls and may contain that pattern,
 *     order them after the library's stream, or run with mocha_set_option "gemm_bf16x3" = 0 and "attention_bf16x3" = 0.
 *     Scope of this statement: ONE builder-written reproduction, observed on the boxes of this build's GPU pool - MI355X (gfx950:sramecc+:xnack-),
 *     kernel driver 6.18.51, ROCm runtime 7.0.2 (HIP 7.0.51831), code objects built with hipcc 7.2.26015, VBIOS 113-M355-01-1K1-020F,
 *     firmware MEC 44 / RLC 43 / SDMA 14 / SMC 04.86.15.106 (profiles/r05/c_versions.txt) - not a vendor erratum and not re-checked
 *     on any other driver / firmware; the canary test re-measures it on whatever box runs the GPU suite.
 */
#ifndef MOCHA_HIP_H
#define MOCHA_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mocha_ctx mocha_ctx;

typedef enum {
    MOCHA_OK = 0,
    MOCHA_ERR_ARG = -1,      /* bad argument / unsupported configuration */
    MOCHA_ERR_HIP = -2,      /* a HIP runtime call failed */
    MOCHA_ERR_STATE = -3,    /* call order violated (weights not finalised, no bank, ...) */
    MOCHA_ERR_WEIGHT = -4    /* unknown weight name or wrong shape */
} mocha_status;

/* The 14 model dimensions Generator.__init__ reads (model.py:18-33, configs/config.yaml:13-31)
 * plus the skeleton layout (0 = 'mocha' 24 joints, 1 = 'mixamo' 22 joints; net/graph.py). */
typedef struct {
    int T, V, C_in, patch, dim;
    int enc_depth, enc_heads, enc_dim_head, enc_mlp;
    int dec_depth, dec_heads, dec_dim_head, dec_mlp;
    int layout;
} mocha_cfg;

/* Generator(cfg) construction, model.py:16-80 / trainer.py:22-23. */
int mocha_create(const mocha_cfg* cfg, int device, mocha_ctx** out);
void mocha_destroy(mocha_ctx* ctx);
const char* mocha_last_error(const mocha_ctx* ctx);   /* ctx may be NULL: last create error */

/* load_state_dict, trainer.py:239-240: one call per state_dict entry, `host` is HOST fp32.
 * Names are exactly the reference's ("mot_embedding.2.blk.gcn.conv.weight", ...).  The graph
 * buffers (A_j, A_b, pool/unpool weights) are accepted and cross-checked against the
 * regenerated constants. */
int mocha_load_weight(mocha_ctx* ctx, const char* name, const float* host, const int64_t* shape, int ndim);
/* After the last mocha_load_weight: checks completeness, repacks weights into kernel layouts. */
int mocha_finalize_weights(mocha_ctx* ctx);

/* Pre-allocate workspaces for up to `max_batch` windows per internal chunk (larger B is
 * processed in chunks).  Optional; called lazily otherwise (not graph-capture safe then). */
int mocha_reserve(mocha_ctx* ctx, int max_batch);

/* model.pos_emb (model.py:40; test_fullframework.py:191): device pointer to (90, 256). */
int mocha_pos_emb(mocha_ctx* ctx, const float** dev_ptr);

/* model.mot_embedding(X), model.py:42-50 (test_fullframework.py:190).
 * X (B,T,V,C_in) -> tokens (B,90,256); add_pos != 0 fuses `+ pos_emb` (test_fullframework.py:191). */
int mocha_embed(mocha_ctx* ctx, const float* X, int B, float* tokens, int add_pos, void* stream);
/* model.encoder(tokens), model.py:53-59, net/transformer.py:90-95 (test_fullframework.py:192). */
int mocha_encoder(mocha_ctx* ctx, const float* tokens, int B, float* encoded, void* stream);
/* mean_variance_norm(encoded.permute(0,2,1)).permute(0,2,1), net/transformer.py:13-20
 * (test_fullframework.py:193).  If cnt_nm is non-NULL also writes cnt_nm = (cnt - cnt_mean) / cnt_std
 * (test_fullframework.py:293,297,442; cnt_mean / cnt_std (90,256) are then required).  cnt may be NULL when only cnt_nm is
 * wanted (at least one of the two must be given). */
int mocha_mvn(mocha_ctx* ctx, const float* encoded, int B, float* cnt,
              const float* cnt_mean, const float* cnt_std, float* cnt_nm, void* stream);
/* Fused demo encode sequence test_fullframework.py:190-193: embed, +pos_emb, encoder and - each when its pointer is non-NULL -
 * the cnt feature and its z-scored copy cnt_nm (as in mocha_mvn: cnt_nm needs cnt_mean / cnt_std, not cnt). */
int mocha_encode(mocha_ctx* ctx, const float* X, int B, float* encoded, float* cnt,
                 const float* cnt_mean, const float* cnt_std, float* cnt_nm, void* stream);
/* model.decoder(src_enc, cha_enc), model.py:62-68, net/transformer.py:90-121 (test_fullframework.py:301,455,465). */
int mocha_decoder(mocha_ctx* ctx, const float* src_enc, const float* cha_enc, int B, float* out, void* stream);
/* AdaIN's style constants of character features (net/transformer.py:98-107: AdaptiveAvgPool1d over the tokens, Linear 256->512,
 * LeakyReLU(0.2), Linear 512->512, of EVERY decoder layer): gb (B, 512 * decoder_depth) = [gamma_0 | beta_0 | gamma_1 | beta_1 | ...] for
 * cha_enc (B, 90, 256) - what model.decoder computes from its second argument before anything else, and what mocha_bank_set caches per
 * bank entry ("bank_dec_cache").  In float64 from a float64 token mean, rounded to fp32 once, unless "style_f64" = 0. */
int mocha_style_constants(mocha_ctx* ctx, const float* cha_enc, int B, float* gb, void* stream);

/* model.to_mot(tokens), model.py:71-80 (test_fullframework.py:302,456,466). */
int mocha_to_mot(mocha_ctx* ctx, const float* tokens, int B, float* Y, void* stream);
/* Generator.forward(src_X, cha_X), model.py:82-106. */
int mocha_forward(mocha_ctx* ctx, const float* src_X, const float* cha_X, int B, float* Y, void* stream);
/* Generator.forward(..., extract_feature=True), model.py:95-98. */
int mocha_forward_features(mocha_ctx* ctx, const float* src_X, const float* cha_X, int B,
                           float* src_enc, float* cha_enc, float* src_cnt, float* cha_cnt, void* stream);

/* Character feature bank = what BallTree(cha_cnt_nm) + cha_encoded hold in the demo
 * (test_fullframework.py:293-294, 298).  cnt_nm (N, 90*256) are the z-scored cnt features,
 * encoded (N, 90, 256) the features gathered for the decoder.  flags bit 0: borrow the
 * caller's buffers instead of copying (they must stay valid and unchanged). */
#define MOCHA_BANK_BORROW 1
/* flags bit 1: additionally keep a bf16 copy of cnt_nm (round-to-nearest-even) and match against it
 * (half the HBM bytes per bank scan; BASELINE configs[2]); indices then agree with the fp32 search
 * except where the two nearest distances differ by less than the bf16 rounding of the bank.
 * fp32 banks of up to 4096 rows additionally keep the packed three-plane image of the centred bank (6 bytes per value, made by
 * mocha_bank_set) that many-query matching multiplies on the bf16 pipe when "gemm_bf16x3" is on - exact planes, the same indices.
 * fp32 banks of at least 4096 rows additionally keep a centred bf16 copy (2 bytes per value) and one residual norm per row
 * (option "scan16", default 1): matching of up to 8 queries scans the copy - the scan is HBM-bound, half the bytes - and
 * re-evaluates on the fp32 rows, exactly, every row the copy's rounding cannot exclude (triangle inequality on the measured
 * residuals): the indices and distances of the fp32 search, not of a bf16 bank. */
#define MOCHA_BANK_BF16 2
int mocha_bank_set(mocha_ctx* ctx, const float* cnt_nm, const float* encoded, int64_t N, int flags, void* stream);
/* tree.query(q, k=1), test_fullframework.py:296,443: exact Euclidean 1-NN of each z-scored
 * query row (Q, 90*256) in the bank.  idx (Q,) int32; dist (Q,) fp32 Euclidean distance to the
 * winner (may be NULL). */
int mocha_match(mocha_ctx* ctx, const float* query_nm, int Q, int32_t* idx, float* dist, void* stream);
/* tree.query(q, k) for k > 1 (scikit-learn BallTree semantics; the reference only ever asks for k = 1, SURVEY.md §8f N4 lists
 * k > 1 as optional): the k nearest rows of every query, exact, distances ascending, ties to the lower row index.
 * idx (Q,k) int32 (-1 where the bank has fewer than k rows), dist (Q,k) fp32 or NULL.  One bank scan per 8 queries - a
 * convenience, not a fast path; allocates its scratch on first use.
 * mocha_bank_gather_blend: soft matching over those neighbours, out (Q,90,256) = sum_j softmax_j(-dist[q][j] / temperature) *
 * bank.encoded[idx[q][j]] (entries with idx < 0 left out; k <= 64, temperature > 0). */
int mocha_match_topk(mocha_ctx* ctx, const float* query_nm, int Q, int k, int32_t* idx, float* dist, void* stream);
int mocha_bank_gather_blend(mocha_ctx* ctx, const int32_t* idx, const float* dist, float temperature, int Q, int k, float* out,
                            void* stream);
/* cha_encoded[frame_index], test_fullframework.py:298,465: out (Q,90,256) = bank.encoded[idx[q]]. */
int mocha_bank_gather(mocha_ctx* ctx, const int32_t* idx, int Q, float* out, void* stream);

/* The NN ("cm_") branch of the demo, batched over all source windows
 * (test_fullframework.py:188-194, 288-302, 438-443, 465-467): encode src, match against the
 * current bank, gather, decode, to_mot.  Y (B,T,V,C_in); idx (B,) may be NULL. */
int mocha_characterize(mocha_ctx* ctx, const float* src_X, int B, const float* cnt_mean, const float* cnt_std,
                       float* Y, int32_t* idx, void* stream);

/* One streamed window per call (BASELINE configs[4]; the demo's per-frame loop test_fullframework.py:438-443, 465-467):
 * mocha_characterize with B = 1, captured into a HIP graph on first use and replayed afterwards (one graph launch
 * instead of ~45 kernel launches).  The graph is keyed on the five buffer pointers, `raw` and mocha_generation(ctx);
 * it is re-captured transparently when any of them changes (new bank, grown workspace, other buffers).  X1 (1,T,V,C_in)
 * [raw != 0: (1,T,V+1,C_in) un-normalised, after mocha_set_pose_norm], Y1 (1,T,V,C_in), idx (1,); all device memory
 * that stays valid between calls.  `stream` may be the null stream (capture runs on an internal stream). */
int mocha_step_graph(mocha_ctx* ctx, const float* X1, const float* cnt_mean, const float* cnt_std, float* Y1, int32_t* idx,
                     int raw, void* stream);
/* The same step on one of the context's LANES (mocha_set_option "lanes" = 1..3; lane 0 is mocha_step_graph).  Every lane has
 * its own workspace set, match scratch and captured graph, so the steps of up to three consecutive windows can be in flight
 * at once, each on its own stream: windows are independent (test_fullframework.py:148-158) and the step alternates a
 * latency-bound chain of small kernels (encode, decode) with an HBM-bound bank scan, so window i + 1's chains run under
 * window i's scan.  The caller gives every lane its own X1 / Y1 / idx buffers and stream; results of one lane are ordered by
 * that stream as usual.  Weights and the bank are shared and read-only. */
int mocha_step_graph_lane(mocha_ctx* ctx, int lane, const float* X1, const float* cnt_mean, const float* cnt_std, float* Y1,
                          int32_t* idx, int raw, void* stream);

/* Multi-GPU set-up: one process per GPU, windows sharded across ranks, the character bank replicated.
 * The reference has no counterpart (trainer.py:45-47 is nn.DataParallel); the only exchange on the path is this one-time
 * bank broadcast over RCCL / xGMI.  RCCL is loaded lazily (dlopen "librccl.so.1"); single-GPU callers never touch it.
 *   mocha_comm_unique_id : rank 0 obtains the 128-byte ncclUniqueId (HOST buffer) and ships it to the other ranks by any
 *                          means (torch.distributed, MPI, a file);
 *   mocha_comm_init      : every rank joins (ncclCommInitRank) — collective;
 *   mocha_bank_broadcast : the root's current bank (mocha_bank_set, N entries) becomes every rank's current bank.
 *                          cnt_nm and encoded travel as scatter (grouped ncclSend/ncclRecv of 1/world each) + in-place
 *                          ncclAllGather, so that every xGMI link of the root carries a share instead of one ring
 *                          neighbour carrying all of it; centroid, row norms and the optional bf16 copy (flags &
 *                          MOCHA_BANK_BF16) are recomputed locally.  `comm` = an ncclComm_t created by the same librccl,
 *                          or NULL for the context's own.  Collective.  Every rank first contributes a 16-byte header
 *                          {entries, bf16?} as it understands the call; the headers are all-gathered and checked by
 *                          every rank (one stream synchronisation), so a root without that bank, a flags mismatch or a
 *                          single rank naming another size fails on ALL ranks instead of hanging some of them; the
 *                          payload is then enqueued on `stream`.
 *   mocha_set_rccl_library: which librccl to resolve (before the first mocha_comm_* call; process-wide).  A process that
 *                          already holds an RCCL - PyTorch wheels bundle their own copy - should name that file, so that one
 *                          RCCL instance serves the process; NULL / never called: "librccl.so.1" from the loader path. */
int mocha_set_rccl_library(const char* path);
int mocha_comm_unique_id(mocha_ctx* ctx, void* id128);
int mocha_comm_init(mocha_ctx* ctx, const void* id128, int nranks, int rank);
int mocha_comm_destroy(mocha_ctx* ctx);
int mocha_bank_broadcast(mocha_ctx* ctx, void* comm, int root, int64_t N, int flags, void* stream);
/* The plan mocha_bank_broadcast follows for one tensor of `count` floats over `world` ranks (host-side, pure; exposed so
 * that the split can be tested without a GPU): out = {offset of this rank's chunk, chunk length, offset of the tail, tail
 * length}.  Chunk r is scattered root -> r and all-gathered; the count % world tail is broadcast whole. */
int mocha_bcast_plan(int64_t count, int world, int rank, int64_t out[4]);
/* What the context's communicator really is, for the bench line and for logs (bench.py's `rccl` record): ranks and this rank
 * as RCCL itself reports them (ncclCommCount / ncclCommUserRank on the communicator mocha_comm_init made - not the caller's
 * arguments), ncclGetVersion's code (major * 10000 + minor * 100 + patch), the file the RCCL entry points were resolved
 * from (dladdr on ncclCommInitRank: a stand-in or a second RCCL cannot pass unnoticed), the context's device ordinal and
 * its PCI bus id (hipDeviceGetPCIBusId).  MOCHA_ERR_STATE without a communicator.  Host-side, no synchronisation. */
typedef struct {
    int nranks, rank, rccl_version, device;
    char pci_bus_id[32];
    char library[512];
} mocha_comm_info_t;
int mocha_comm_info(mocha_ctx* ctx, mocha_comm_info_t* out);

/* Pose normalisation of the demo fused into the path: the four norm.npz
 * arrays of the reference (test_fullframework.py:64-71), HOST fp32, (V+1)*C_in each with the root bone
 * first.  Afterwards the *_raw entry points take un-normalised poses WITH the root bone,
 * X_raw (B,T,V+1,C_in), apply X = (X[:,:,1:] - X_mean[:,:,1:]) / X_std[:,:,1:] on load
 * (test_fullframework.py:186,269) and return Y * Y_std[0,:,1:] + Y_mean[0,:,1:] (test_fullframework.py:303,457). */
int mocha_set_pose_norm(mocha_ctx* ctx, const float* x_mean, const float* x_std, const float* y_mean, const float* y_std);
int mocha_encode_raw(mocha_ctx* ctx, const float* X_raw, int B, float* encoded, float* cnt,
                     const float* cnt_mean, const float* cnt_std, float* cnt_nm, void* stream);
int mocha_characterize_raw(mocha_ctx* ctx, const float* src_X_raw, int B, const float* cnt_mean, const float* cnt_std,
                           float* Y_denorm, int32_t* idx, void* stream);

/* The demo pair in one pass: the character clip (B_cha windows) becomes the bank, the B_src source windows are characterized
 * against it — mocha_encode(cha) + mocha_bank_set + mocha_characterize(src) with both clips sharing every launch of
 * mot_embedding / encoder / cnt (test_fullframework.py:188-194, 271-277 for the two clips; :293-296, 438-443, 465-467).
 * The bank is transient (it lives in the workspace during the call); the context's own bank is left untouched.  Optional
 * outputs cha_encoded / cha_cnt_nm (B_cha,90,256) receive the bank for later mocha_bank_set; idx (B_src) the matches.
 * B_src + B_cha must fit the workspace limit (default 1280 windows, mocha_reserve).  *_raw: as mocha_characterize_raw. */
int mocha_characterize_pair(mocha_ctx* ctx, const float* src_X, int B_src, const float* cha_X, int B_cha, const float* cnt_mean,
                            const float* cnt_std, float* Y, int32_t* idx, float* cha_encoded, float* cha_cnt_nm, void* stream);
int mocha_characterize_pair_raw(mocha_ctx* ctx, const float* src_X_raw, int B_src, const float* cha_X_raw, int B_cha,
                                const float* cnt_mean, const float* cnt_std, float* Y, int32_t* idx, float* cha_encoded,
                                float* cha_cnt_nm, void* stream);

/* CVAE character-feature sampler: CVAE.sample(c) of model_CVAE.py:44-46, i.e.
 * PriorNet (:49-92) then Decoder (:138-165).  Weights by their reference state_dict names
 * ("prior_net.encoder.layers.0.self_attn.in_proj_weight", ...; test_fullframework.py:52-58); the
 * posterior "encoder.*" entries and the pos_encoder.pe buffers of a checkpoint are accepted and ignored.
 * cond (B,180,256) -> out (B,90,256).  eps (B,256) is the reparameterisation noise
 * z = mu + eps*exp(0.5*logvar) (:81-87); eps == NULL is deterministic=True (z = mu).  mu/logvar (B,256) optional. */
int mocha_cvae_load_weight(mocha_ctx* ctx, const char* name, const float* host, const int64_t* shape, int ndim);
int mocha_cvae_finalize(mocha_ctx* ctx);
int mocha_cvae_sample(mocha_ctx* ctx, const float* cond, int B, float* out, float* mu, float* logvar, const float* eps, void* stream);

/* Conditioning glue of the demo's CVAE ("Ours") branch, test_fullframework.py:446-449, all (.., 90, 256):
 *   cond (B,180,256) = cat[(src_cnt - src_mean)/src_std , (prev_cha - cha_mean)/cha_std]  (token axis)
 *   out = x * std + mean                       (curr_cha_encoded = vae_output * cha_encoded_std + cha_encoded_mean) */
int mocha_cvae_condition(mocha_ctx* ctx, const float* src_cnt, const float* src_mean, const float* src_std, const float* prev_cha,
                         const float* cha_mean, const float* cha_std, int B, float* cond, void* stream);
int mocha_scale_shift(mocha_ctx* ctx, const float* x, const float* mean, const float* std_, int B, float* out, void* stream);

/* Window featurisation of the demo (test_fullframework.py:141-185): local bone features
 * of B windows — Yrot (B,T,V+1,4) quaternions (w,x,y,z), Ypos / Yvel / Yang (B,T,V+1,3), root bone first, as
 * process_data produces them (:126-139) — to the un-normalised features X_raw (B,T,V+1,15) that the *_raw
 * entry points consume: FK with velocities, re-rooting on each window's last frame, root-relative
 * position | rotation-matrix xy | velocity | angular velocity. */
int mocha_featurize(mocha_ctx* ctx, const float* Yrot, const float* Ypos, const float* Yvel, const float* Yang, int B,
                    float* X_raw, void* stream);

/* Post-processing of decoded windows.
 * mocha_pose_heads: Y (B,60,V,15) de-normalised -> heads (B,V,13) = [pos 3 | quat wxyz 4 | vel 3 | ang 3] of the last
 *   frame (test_fullframework.py:304-308: slices + quat.from_xform_xy) and speed (B) = mean_t |Y[t,0,9:12]| (:338).
 * mocha_postprocess: the sequential frame loop of the demo for n_clips independent clips of n_frames windows each
 *   (:338-437 first frame, :492-632 after it): root integration from the source's root-local velocities, position
 *   blending, foot-lock state machine (motion/Inertialization.py:300-377) and two-bone IK (motion/quat.py:295-343);
 *   optionally the root merge + Euler channels of the BVH writer (:677-681, 697).  State is float64 as in the
 *   reference.  heads (clips,frames,V,13), speed/src_speed (clips,frames), src_rvel/src_rang (clips,frames,3),
 *   contact (clips,frames,n_contact) uint8 -> pos (clips,frames,V+1,3), rot / ik_rot (clips,frames,V+1,4) wxyz,
 *   bvh_pos / bvh_euler (clips,frames,V,3; degrees; both NULL to skip).  cfg NULL = the demo's constants (:104-114). */
typedef struct mocha_post_cfg {
    double dt, ik_max_length_buffer, ik_foot_height, ik_unlock_radius, ik_blending_halflife;
    int ik_enabled, n_contact;
    int contact_bones[4];      /* toe bones, indices in the (V+1)-bone skeleton with the root bone in front */
    int blend_enabled;         /* 1 (default): positions blended with the previous frame's (:537, 627).  0 together with
                                * ik_enabled = 0 is the demo's context-matching "cm_" stream: decoded poses as they are, only
                                * the root integrated (test_fullframework.py:512-527, 637-641) */
} mocha_post_cfg;
void mocha_post_cfg_default(mocha_post_cfg* cfg);
int mocha_pose_heads(mocha_ctx* ctx, const float* Y, int B, float* heads, float* speed, void* stream);
int mocha_postprocess(mocha_ctx* ctx, const mocha_post_cfg* cfg, const float* heads, const float* speed, const float* src_rvel,
                      const float* src_rang, const float* src_speed, const unsigned char* contact, int n_clips, int n_frames,
                      double* pos, double* rot, double* ik_rot, double* bvh_pos, double* bvh_euler, void* stream);

/* Bank build statistics: cnt_mean, cnt_std = np.mean(cnt, 0), np.std(cnt, 0) over the N bank entries
 * (compute_cnt_norm.py:174-175; population std), x (N, 90*256) -> mean, std (90*256). */
int mocha_column_stats(mocha_ctx* ctx, const float* x, int64_t N, float* mean, float* std_, void* stream);

/* Runtime options.  "dual_stream" (default 0): batches of at least "dual_min" (default 128) windows are split in two
 * halves that run concurrently on the caller's stream and on an internal stream (forked / joined with events, so
 * the call keeps stream semantics and stays graph-capturable); the two kernel chains fill each other's prologue /
 * epilogue / tail bubbles (+5 % on the demo step).  Results are unchanged except for the kernel choice per half.
 * "fold_decoder" (default 1): the decoder's key and value projections are folded into its query and output weights at
 * mocha_finalize_weights (possible because decoder_dim_head == dim; exact algebra, differences are fp32 rounding only),
 * which removes two of the four projection GEMMs per decoder layer; 0 runs the four projections as written in
 * net/transformer.py:62-76.
 * "gemm_bf16x3" (default 1): the large GEMM launches (every nn.Linear / conv-as-GEMM of the path whose batch fills the chip
 * and whose output width is a multiple of 64) run on the bf16 matrix pipe with both fp32 operands carried as three bf16
 * planes and six MFMA passes per product, fp32 accumulation (gemm_x3.hip): every bf16 x bf16 product is exact in fp32 and
 * the dropped cross terms are below 2^-26 of a product, so the result is as accurate as an fp32 FMA chain (measured against
 * float64: slightly more accurate than the exact-f32 MFMA kernel, tests/test_gemm_engines.py).  0 = every GEMM on
 * v_mfma_f32_32x32x2_f32 (gemm_f32.hip).
 * "attention_bf16x3" (default 1): the Generator's attention (net/transformer.py:65-76; 90 tokens, head dim 128 / 256) computes
 * QK^T and PV the same way - K, Q, V and the softmax probabilities as three bf16 planes each, six MFMA passes per product,
 * fp32 accumulation and an fp32 softmax (attention_x3.hip); 0 = v_mfma_f32_32x32x2_f32 (attention.hip).  The CVAE sampler's
 * attention (head dim 64) always uses attention.hip.
 * "attention_split_max" (default 192): up to this many (window, head) pairs the head-dim-256 attention
 * (the decoder's) gives each pair twelve waves - four groups contract a quarter of the head dim each, the partial score tiles are
 * summed through LDS in a fixed order, each group then writes a quarter of the output columns - because a one-window launch is
 * otherwise a serial chain of twelve staging steps per wave.  Same operands and products; the scores are summed in a different
 * association than the three-wave kernel's, so results agree to fp32 rounding.  0 = always the three-wave kernel.
 * "fold_joint" (default 1): the embedding joint block's 1x1 gcn conv folded into its k = 5 temporal conv at
 * mocha_finalize_weights (net/blocks.py:126-134 applies them back to back with nothing in between: one linear map, K = 5 x 192);
 * 0 runs the two convolutions as two GEMMs.  Differences are fp32 rounding only.
 * "lanes" (default 1, 1..3): workspace sets / captured graphs for mocha_step_graph_lane; changing it re-plans the workspaces.
 * "scan16" (default 1): see MOCHA_BANK_BF16 above - fp32 banks of >= 4096 rows are scanned through a centred bf16 copy with an
 * exact fp32 re-rank when at most 8 queries are matched; takes effect at the next mocha_bank_set; 0 scans the fp32 rows.
 * Round 4 (INTEGRATION.md section 6 has the table; PER CONTEXT since ABI 5 - round 4 kept five of them in process-wide variables):
 * "embed_sums" (default 1): mot_embedding's first stage and the joint
 * block's five-tap 4-frame sums in one kernel whose frame rows stay in LDS (0: two kernels, bit-identical); "embed_front_max_wgs"
 * (default 512): that kernel's persistent grid; "inorm_split_max" (default: every batch): windows up to
 * which the instance norms spread a window's channels over four workgroups (bit-identical either way); "gemm_persistent" (default 768
 * workgroups; 0 = off) and "gemm_persistent_max_n" (default 512): the plane GEMM's persistent instance, bit-identical; "select2"
 * (default 1) / "match_planes" (default 1; 2 = two bf16 query planes in the bf16 coarse pass): the many-query selection with
 * producer-side row statistics - the same exact result; "attention_kv", "attention_kv_pairs" (default 0): decoder attention from
 * pre-split key / value images, bit-identical; "bank_tiled" (default 0): bf16 banks, a tiled image for the many-query coarse pass;
 * "dual_min" (default 128): smallest batch that "dual_stream" splits over two streams (re-plans the workspaces).
 * "fold_upsample" (default 1): to_mot's k = 5 temporal conv over the nearest-x4-upsampled frames (model.py:74, net/blocks.py:112-118) as a
 * 3-tap conv over the 15 source frames with per-phase summed weights (exact algebra; differences are fp32 rounding of the weight sums);
 * "upsample_split_min" (default 256): windows from which it runs as two 2-tap launches.
 * Round 5 (ABI 5):
 * "adain_closed_form" (default 1): the decoder's AdaIN -> instance norm pair (net/transformer.py:108-113 then :49-56) from ONE set of
 * token statistics: with m, s the mean / unbiased std of x, AdaIN(x) = (1+g)(x-m)/(s+eps) + b has token mean exactly b and std
 * |1+g| s/(s+eps), so IN(AdaIN(x)) = (1+g)(x-m) / (|1+g| s + eps (s+eps)) - the same function in real arithmetic, without the
 * cancellation against b that costs the literal fp32 order (and the reference) the digits of every channel whose 1+g is small.
 * 0 = the literal two-pass order (A/B: tools/structured_matrix.py).
 * "style_f64" (default 1): the AdaIN style MLP (net/transformer.py:100-107; mean over tokens -> Linear -> LeakyReLU -> Linear) in
 * float64 from a float64 token mean, gamma / beta rounded to fp32 once; 0 = the fp32 GEMM engines.
 * "bank_dec_cache" (default 1): mocha_bank_set also computes what the decoder derives from a bank ENTRY alone - IN(entry) (the folded
 * decoder's keys) and every layer's gamma / beta - (+ 92 KB + 2 KB x decoder depth per entry), and mocha_characterize /
 * mocha_step_graph read them in place through frame_index instead of recomputing them per call from a gathered copy
 * (needs "style_f64", "fold_decoder", decoder_dim_head == dim, "attention_bf16x3", no "attention_kv"; otherwise, and with 0, the
 * per-call flow).  Takes effect at the next mocha_bank_set.
 * "match_pass" (default 0 = round 4's mocha_match_gemm_bf16_dma): 1 / 2 = the bf16 bank's coarse pass for up to "match_pass_max_q"
 * (default 256) queries as mocha_match_pass256 (match_pass.hip: 128 x 256 tiles, bank operands straight into registers, K split 16) on
 * the row-major bank / on an operand-order image built at the first such call (+ 2 B per bank value).  Same products summed over other K
 * slices: the selection's exact re-evaluation makes the answer the same.  Measured no faster at 128 queries, 13 us faster at 256 from the
 * image.  "match_pass_variant": that kernel's prefetch depth / non-temporal bit (diagnostic).
 * "match_nt" (default 1): the round-4 kernel's bank loads carry the non-temporal hint when a launch reads the bank once (Q <= 128).
 * "pair_overlap" (default 1): mocha_characterize_pair computes the transient bank's decoder constants on the context's internal stream
 * beside the matching chain (forked / joined with events; bit-identical, -0.6 % of the demo step).
 * "match_fold" (default 0; a measured negative kept reproducible): the bf16 many-query coarse pass's last-arriving
 * K-slab workgroup per tile adds the slabs up into slab 0 and the selection reads one slab; same indices, the pass 53 us slower.
 * "scan8" (default 0): mocha_bank_set also keeps the centred rows of an fp32 bank of >= 4 096 entries as biased bytes with a per-row scale and a
 * measured residual bound (+ N x 23 040 B); mocha_match with <= 4 queries then scans 1 B per value and re-evaluates exactly, on the fp32 rows,
 * every row the byte image cannot exclude - the result of the exact fp32 search - and falls back, by itself and on the device, to the bf16 scan
 * for banks / queries where the image excludes too little (independent N(0, 1) rows).  Takes effect at the next mocha_bank_set.
 * "gemm_x3r_min_n" (default 0 = never): K = 256 plane-GEMM launches of at least 8 192 rows and at least this many columns (a multiple of 256)
 * take the instance that keeps a wave's 32 activation rows resident in registers as planes for all n-tiles (gemm_x3r.hip, round 6):
 * bit-identical to the tiled instances; measured SLOWER inside the step (enc.qkv 404 -> 422 us, dec.q 144 -> 208, ff1 127 -> 166:
 * one wave per SIMD has nobody to hide behind; profiles/r06), so it stays an option.
 * "gemm_tile64_below" (default 0): mid-size plane-GEMM launches of 128-multiple width with fewer 64 x 128 tiles than this take 64 x 64
 * tiles (measured no gain); widths that are multiples of 64 only always do at mid size.
 * "gemm_f16x2" (default 0): the encoder's, decoder's and to_mot's batch-size GEMMs as TWO fp16 planes per operand and THREE
 * v_mfma_f32_32x32x16_f16 passes per product instead of three bf16 planes and six passes (gemm_h2.hip): x = fp16(S x) + fp16(residual)
 * carries 22 bits, a b ~ a0 b0 + a0 b1 + a1 b0, fp32 accumulation.  Per product 2^-22 instead of an fp32 multiply's 2^-24; a whole dot
 * product comes out CLOSER to float64 than on either fp32 engine (the accumulator's roundings dominate; tests/test_gemm_f16x2.py) at
 * 0.65-0.75 of the time.  Power-of-two scales: per weight row at pack time; per WINDOW for the activations, from bounds the producing
 * launch leaves in device memory (or that the arithmetic implies) - a window's result does not depend on the rest of the batch; rows more
 * than five decades below their window's largest magnitude keep fewer digits than fp32 would.  Launches without a bound (the raw-pose
 * embedding, calls of up to four windows) stay on the bf16 planes; the attention and the matcher always do.  The current bank's per-entry
 * bounds are taken at the next mocha_bank_set.
 * Every option that changes which kernels a step launches bumps mocha_generation(ctx). */
int mocha_set_option(mocha_ctx* ctx, const char* name, int value);

/* y (M,N) = x (M,K) · w (N,K)^T + bias (N, may be NULL): nn.Linear (net/transformer.py:28-32, 57-61) as a stand-alone
 * operator on device pointers, for tests and tooling that want one of the two GEMM engines directly.
 * engine 0: the kernel the path would pick for this shape under the current options; 1: exact-f32 MFMA; 2: bf16 x 3 planes;
 * 3: fp16 x 2 planes / three passes (option "gemm_f16x2"'s engine; the activation bound is measured on every call; N % 64, K % 32)
 * (MOCHA_ERR_ARG if the shape is outside that engine: K % 32, N % 128, at least 768 row tiles ... see gemm_x3_supports).
 * With engine 2 (or 0 resolving to it) w is packed into the engine's image on every call (w may change between calls);
 * the call synchronises the stream once to free that image.  K % 32 == 0. */
int mocha_linear(mocha_ctx* ctx, const float* x, const float* w, const float* bias, float* y, int64_t M, int N, int K, int engine,
                 void* stream);

/* Introspection for tests and tooling. */
int mocha_abi_version(void);
/* Build provenance: the HIP toolchain libmocha_hip.so was compiled with ("hipcc HIP 7.2.26015-fc0010cf6a gfx950", static
 * string) and the HIP runtime version of the process that loaded it (hipRuntimeGetVersion; no device needed).  The built
 * library ships to the GPU box, whose ROCm may be older than the build container's: smoke() prints both. */
const char* mocha_build_info(void);
int mocha_runtime_version(void);
/* Monotonic counter, bumped whenever the context replaces a device buffer a captured graph may hold (workspaces, match
 * scratch, CVAE workspace) or its current bank changes.  Host-side; no synchronisation. */
int64_t mocha_generation(const mocha_ctx* ctx);
int mocha_graph_constants(mocha_ctx* ctx, float* A_j /*3*V*V host*/, float* A_b /*2*6*6 host*/,
                          float* pool /*V*6 host*/, float* unpool /*6*V host*/);

/* Copies of the current bank's rows into caller buffers, (N, 90*256) fp32 each, either may be NULL: what a rank that RECEIVED its bank
 * (mocha_bank_broadcast fills context-owned buffers) hands to further contexts of its process (mocha_bank_set ... MOCHA_BANK_BORROW on each:
 * BatchPipeline).  Enqueued on `stream`. */
int mocha_bank_export(mocha_ctx* ctx, float* cnt_nm, float* encoded, void* stream);

/* The current bank as the context holds it (device pointers, nothing is copied): the rows it matches against and gathers
 * from, and what the library derived from them - the centroid (90*256), the squared norms of the centred rows (N), the
 * centred bf16 copy (NULL for an fp32 bank).  Any out pointer may be NULL.  For tests that compare the ranks of a
 * mocha_bank_broadcast bit for bit. */
/* Option "scan8" (round 6; the matcher's one-byte first stage, match_scan8.hip): state[0] = 1 if the current bank carries the byte image,
 * state[1] = the stage's sticky mode word of workspace set `set` read back from the device (0: calls scan the byte image; 1: the refine
 * found the image's bounds useless for this bank / these queries and calls scan the bf16 copy; -1: no scratch yet).  Synchronises the device.
 * Replaces nothing in the reference (its BallTree has no such stage): introspection for tests and tooling. */
int mocha_scan_byte_state(mocha_ctx* ctx, int set, int32_t* state /*2*/, void* stream);
int mocha_bank_view(mocha_ctx* ctx, const float** cnt_nm, const float** encoded, const float** centroid, const float** row_norm2,
                    const void** cnt_bf16, int64_t* N);

/* Measurement support (bench.py roofline leg): between start and stop every kernel launch is
 * bracketed by a HIP event pair on its launch stream.  stop synchronises those events and
 * writes a JSON object {"kernels": {symbol: {launches, ms, flops, bytes}}, "sites": {...}} of
 * per-kernel totals (algorithmic flops/bytes of the launches) into `json` (capacity `cap`). */
int mocha_profile_start(mocha_ctx* ctx);
int mocha_profile_stop(mocha_ctx* ctx, char* json, int64_t cap);

#ifdef __cplusplus
}
#endif
#endif /* MOCHA_HIP_H */
