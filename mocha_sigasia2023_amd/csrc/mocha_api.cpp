// Host side of libmocha_hip.so: context, weight loading / repacking, workspace management and
// the C-ABI entry points of include/mocha_hip.h.  Every function only enqueues work on the
// caller's stream.  Reference citations are relative to the reference repository root.
#include "../../include/mocha_hip.h"
#include "kernels.h"

#include <rccl/rccl.h>      // types only: the RCCL entry points are resolved at run time (see Rccl below)
#include <dlfcn.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <deque>
#include <map>
#include <tuple>
#include <string>
#include <vector>

using namespace mocha;

namespace {

thread_local std::string g_create_error;

struct HostTensor {
    std::vector<float> data;
    std::vector<int64_t> shape;
};

// ------------------------------------------------------------------ skeleton constants (row a14)
// closed form of net/graph.py:116-131,290-312, computed in double
struct Skeleton {
    int V = 0;
    std::vector<int> parents, part_of;
    std::vector<double> A_j;   // [3][V][V]
    std::vector<double> A_b;   // [2][6][6]
    std::vector<double> pool;  // [V][6]
    std::vector<double> unpool;// [6][V]
};

static std::vector<double> distance_adjacency(const std::vector<int>& parents, int max_hop) {
    const int n = (int)parents.size();
    std::vector<std::vector<int>> nbr(n);
    for (int i = 0; i < n; ++i)
        if (parents[i] >= 0) { nbr[i].push_back(parents[i]); nbr[parents[i]].push_back(i); }
    std::vector<int> hop(n * n, 1 << 20);
    for (int s = 0; s < n; ++s) {
        std::deque<std::pair<int, int>> q;
        std::vector<char> seen(n, 0);
        hop[s * n + s] = 0; seen[s] = 1; q.push_back({s, 0});
        while (!q.empty()) {
            auto [v, d] = q.front(); q.pop_front();
            if (d == max_hop) continue;
            for (int u : nbr[v]) if (!seen[u]) { seen[u] = 1; hop[s * n + u] = d + 1; q.push_back({u, d + 1}); }
        }
    }
    std::vector<double> A((size_t)(max_hop + 1) * n * n, 0.0);
    for (int w = 0; w < n; ++w) {
        int cnt = 0;
        for (int v = 0; v < n; ++v) cnt += hop[v * n + w] <= max_hop;
        for (int v = 0; v < n; ++v) {
            const int h = hop[v * n + w];
            if (h <= max_hop) A[((size_t)h * n + v) * n + w] = 1.0 / (double)cnt;
        }
    }
    return A;
}

static bool make_skeleton(int layout, Skeleton& sk) {
    // net/graph.py:65-79,401-417 ('mocha'); :18-31,329-345 ('mixamo')
    static const int mocha_par[24] = {-1, 0, 1, 2, 3, 0, 5, 6, 7, 8, 9, 10, 11, 8, 13, 14, 8, 16, 17, 18, 0, 20, 21, 22};
    static const int mocha_part[24] = {0, 1, 1, 1, 1, 0, 0, 0, 0, 2, 2, 2, 2, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5};
    static const int mix_par[22] = {-1, 0, 1, 2, 3, 4, 3, 6, 7, 8, 3, 10, 11, 12, 0, 14, 15, 16, 0, 18, 19, 20};
    static const int mix_part[22] = {0, 0, 0, 0, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5};
    if (layout == 0) { sk.V = 24; sk.parents.assign(mocha_par, mocha_par + 24); sk.part_of.assign(mocha_part, mocha_part + 24); }
    else if (layout == 1) { sk.V = 22; sk.parents.assign(mix_par, mix_par + 22); sk.part_of.assign(mix_part, mix_part + 22); }
    else return false;
    sk.A_j = distance_adjacency(sk.parents, 2);
    sk.A_b = distance_adjacency({-1, 0, 0, 0, 0, 0}, 1);
    int cnt[6] = {0, 0, 0, 0, 0, 0};
    for (int v = 0; v < sk.V; ++v) cnt[sk.part_of[v]]++;
    sk.pool.assign((size_t)sk.V * 6, 0.0);
    sk.unpool.assign((size_t)6 * sk.V, 0.0);
    for (int v = 0; v < sk.V; ++v) {
        // the reference forms 1/|part| in float32 (net/graph.py:459-461): keep that rounding
        sk.pool[(size_t)v * 6 + sk.part_of[v]] = (double)(1.0f / (float)cnt[sk.part_of[v]]);
        sk.unpool[(size_t)sk.part_of[v] * sk.V + v] = 1.0;
    }
    return true;
}

struct DevBuf {
    float* p = nullptr;
    size_t n = 0;
};

}  // namespace

struct mocha_ctx {
    mocha_cfg cfg{};
    int device = 0;
    std::string err;
    Skeleton sk;
    std::map<std::string, HostTensor> host_w;
    std::map<std::string, std::vector<int64_t>> expect;
    bool finalized = false;

    std::vector<float*> owned;                 // every hipMalloc'ed pointer
    std::map<std::string, float*> w;           // repacked device weights by short name
    std::map<std::string, size_t> wsize, cwsize;   // element counts of w / cw entries (re-loading reuses the allocation)
    // Bumped whenever a device buffer that a captured graph may have baked in is replaced (workspaces, match scratch,
    // CVAE workspace) or the current bank changes: graph holders compare mocha_generation() before replaying.
    int64_t generation = 1;
    int ntok = 90, nT15 = 15, dim = 256, d4 = 64;

    // workspaces: sized for `chunk` windows; larger batches are processed chunk by chunk
    int chunk = 0;
    int max_chunk = 1280;
    // Two workspace sets: large batches are split in two halves that run on two HIP streams (the caller's and
    // `aux`), so that the prologue / epilogue / tail of one half's kernels overlaps the other half's MFMA phases
    // (+5..10 % measured); `cur` selects the set the pipelines below write to.
    static constexpr int MAX_SETS = 3;
    std::map<std::string, DevBuf> wss[MAX_SETS];
    int cur = 0;
    // Lanes: up to three per-window steps in flight at once (mocha_step_graph_lane; BASELINE configs[4] pipelined).  Lane k
    // owns workspace set k (sets beyond the first are sized for a handful of windows unless dual_stream needs set 1 whole) and
    // its own captured graph; `lane` is the set the single-stream pipelines below write to.
    int lanes = 1, lane = 0;
    bool dual_stream = false; int dual_min = 128;     // opt-in: mocha_set_option(ctx, "dual_stream", 1) or MOCHA_DUAL_STREAM=1
    bool dual_ready = false;                          // ensure_ws allocated set 1 at full chunk size (a lane's set 1 holds 8 windows only)
    int attn_split_max = 192;                         // (window, head) pairs up to which the twelve-wave decoder attention is launched
    // folded decoder, larger batches: attention from pre-split key / value images (attention_kv.hip).  OFF by default: measured at the demo
    // step's 585 windows 197 us (twelve waves per window: three rounds of 256 workgroups) / 168 us (six waves per head pair) against 164 us
    // for mocha_attention_x3<256>, plus 18 us more in the instance norm that writes the images; it wins only where the windows fit one
    // round of CUs (256 windows: 75 against 89 us) - profiles/r04/b_attn_kv_ab.txt
    bool attn_kv = false;
    bool attn_kv_pairs = false;                       // ... with a six-wave workgroup per head pair instead of twelve waves per window (diagnostic)
    hipStream_t aux = nullptr; hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    DevBuf match_S[MAX_SETS];
    int32_t* idx_ws[MAX_SETS] = {nullptr, nullptr, nullptr}; size_t idx_ws_n = 0;

    // bank
    const float* bank_cnt = nullptr;
    const float* bank_enc = nullptr;
    float* bank_cnt_own = nullptr; float* bank_enc_own = nullptr; size_t bank_cap = 0;
    float* bank_norm = nullptr; size_t bank_norm_cap = 0;
    float* bank_center = nullptr;                           // centroid of the matching bank (90*256), see do_match
    float* center_scratch = nullptr;                        // fp64 partial column sums of launch_column_mean
    DevBuf match_qc[MAX_SETS];                                     // queries minus the centroid
    float* pair_norm = nullptr; size_t pair_norm_cap = 0;
    float* pair_center = nullptr;                           // row norms / centroid of the transient bank of mocha_characterize_pair
    unsigned short* pair_x3 = nullptr; size_t pair_x3_cap = 0;     // ... and its packed plane image (never the user's bank_x3)
    int64_t bank_N = 0;
    // Round 5: what the decoder derives from a bank ENTRY alone - IN(cha) (the folded decoder's keys, net/transformer.py:49-56) and every
    // layer's AdaIN gamma / beta (the style MLP of the entry's token mean, :98-107) - is computed once at mocha_bank_set, in float64 for
    // the MLP, and read through frame_index by the decoder (adain: gb_idx, attention: kv_idx) instead of being recomputed per step from
    // a gathered copy: + 92 KB + 4 KB per entry.  Option "bank_dec_cache" (0: recompute per call, the round-4 flow).
    bool bank_dec_cache = true;
    float* bank_kin = nullptr; float* bank_gb = nullptr; size_t bank_dec_cap = 0; bool bank_dec_valid = false;
    double* style_scratch = nullptr; size_t style_scratch_rows = 0;     // float64 token means + hidden activations of the style MLP, bank build
    // launch tuning, per context (round 4 kept these as process-wide globals: a second context, or another host thread, changed them underfoot)
    int inorm_split_max = 1 << 30, embed_max_wgs = 512, gemm_persistent = 768, gemm_persistent_max_n = 512; bool embed_sums = true;
    int gemm_x3r_min_n = 0;            // plane GEMM with the activations resident in registers (gemm_x3r.hip) for K = 256 launches at least this wide (0 = never); option "gemm_x3r_min_n"
    int gemm_tile64_below = 0;         // plane GEMM: 64 x 64 tiles for mid-size 128-multiple launches with fewer 64 x 128 tiles than this (measured: no gain; gemm_x3.hip)
    bool pair_overlap = true;          // characterize_pair: the transient bank's decoder constants on the internal stream, beside the matching
    bool adain_closed = true;          // mocha_adain: qin from the first statistics in closed form (pointwise.hip); 0 = the literal two-pass order
    bool style_f64 = true;             // the style MLP in float64 (mocha_linear_f64); 0 = the fp32 GEMM engines
    std::map<std::string, double*> w64;                     // float64 copies of the style MLP's weights
    // CVAE sampler (row N1): weights under "cvae.<reference key>", workspace for cvae_B conditions
    std::map<std::string, std::vector<int64_t>> cvae_expect;
    std::map<std::string, HostTensor> cvae_host;
    std::map<std::string, float*> cw;
    bool cvae_ready = false;
    int cvae_depth = 2, cvae_heads = 4, cvae_nc = 180, cvae_nq = 90;
    std::map<std::string, DevBuf> cws; int cvae_B = 0;
    // fp32 GEMMs on the bf16 matrix pipe (gemm_x3.hip): packed three-plane images of the weights, made on first use
    bool gemm_x3 = true;
    bool attn_x3 = true;               // the Generator's attention as plane products on the bf16 pipe (attention_x3.hip)
    std::map<std::tuple<const float*, int, int>, unsigned short*> x3w;
    // fp32 GEMMs on the fp16 matrix pipe with two planes / three passes (gemm_h2.hip; option "gemm_f16x2", default off): packed images and
    // per-row inverse scales of the weights (made on first use), and the PER-WINDOW activation bounds the launches scale by: vectors of
    // AMAX_WIN floats in `amax` (entry w = window w of the chunk), AMAX_SLOTS per workspace set and stage (0 encoder, 1 decoder, 2 to_mot,
    // 3 embedding; a stage zeroes the slots it can take when it starts, so a later stage may read an earlier one's), `amax_ok` = the slot was
    // really written by a launch of that engine or by mocha_absmax (host-side bookkeeping)
    bool gemm_h2 = false;
    struct H2Img { unsigned short* img; float* w_inv; };
    std::map<std::tuple<const float*, int, int>, H2Img> h2w;
    static constexpr int AMAX_SLOTS = 40, AMAX_STAGES = 4;             // stages: 0 encoder, 1 decoder, 2 to_mot, 3 embedding
    static constexpr int AMAX_USED[4] = {2 + 4 * 8, 1 + 3 * 8, 5, 5};      // slots a stage can take (depth <= 8): what its start zeroes
    static constexpr int AMAX_WIN = 4096;                             // windows per chunk the bound vectors cover (larger chunks: the option stays off for them)
    float* amax = nullptr;                                            // MAX_SETS x AMAX_STAGES x AMAX_SLOTS vectors of AMAX_WIN floats, then amax_in
    std::vector<char> amax_ok = std::vector<char>(MAX_SETS * AMAX_STAGES * AMAX_SLOTS, 0);
    int amax_stage = 0, amax_next = 0; bool amax_idle = true;
    float* amax_in = nullptr;                                         // constant vector: the bound of an instance-normalised token, (n - 1) / sqrt(n) < 9.5 for 90 tokens
    float* amax_bank = nullptr; size_t amax_bank_cap = 0; bool amax_bank_ok = false;     // per entry of the current bank: the largest magnitude of its encoded rows (mocha_bank_set)
    const float* amax_enc_of[MAX_SETS] = {nullptr, nullptr, nullptr}; float* amax_enc[MAX_SETS] = {nullptr, nullptr, nullptr};   // encoder output pointer -> its slot
    const float* amax_dec_of[MAX_SETS] = {nullptr, nullptr, nullptr}; float* amax_dec[MAX_SETS] = {nullptr, nullptr, nullptr};   // decoder output pointer -> its slot
    const float* amax_tok_of[MAX_SETS] = {nullptr, nullptr, nullptr}; float* amax_tok[MAX_SETS] = {nullptr, nullptr, nullptr};   // embedding output pointer -> its slot
    float emb_l1 = 0.f, emb_bmax = 0.f;                               // largest row L1 norm / bias magnitude of the input 1x1 conv (model.py:44): |conv(x)| <= emb_l1 max|x| + emb_bmax
    bool fold_decoder = true;          // decoder key / value projections folded into the query / output weights
    int upsample_split_min = 256;      // windows from which that conv runs as two 2-tap launches (no zero weight blocks)
    bool fold_upsample = true;         // to_mot: the k=5 temporal conv over the x4-upsampled frames as a 3-tap conv over the SOURCE frames with per-phase summed weights
    bool fold_joint = true;            // embedding joint block: gcn 1x1 conv folded into the k=5 temporal conv (one K = 960 GEMM)
    int* bone_parents = nullptr;       // device: parents of the (V+1)-bone skeleton with the root bone in front
    float* pose_norm = nullptr;        // [x_mean | x_std | y_mean | y_std], (V+1)*C_in each (norm.npz of the reference)
    void* bank_bf16 = nullptr; size_t bank_bf16_cap = 0; bool bank_is_bf16 = false;
    // the bf16 bank once more as the many-query coarse pass's own tiled image (contiguous 16 KB blocks), made at the first such match
    void* bank_tiled = nullptr; size_t bank_tiled_cap = 0; bool bank_tiled_valid = false;
    bool use_tiled = false;            // option "bank_tiled": -4 % on the cold pass for 2 x the bf16 copy's memory (profiles/r04/c_tiled_loader_ab.txt): off
    // fp32 banks of up to X3_BANK_MAX rows also keep the plane engine's packed image of the centred bank (many-query matching)
    unsigned short* bank_x3 = nullptr; size_t bank_x3_cap = 0; bool bank_x3_valid = false;
    unsigned long long* best_ws[MAX_SETS] = {nullptr, nullptr, nullptr}; size_t best_ws_n[MAX_SETS] = {0, 0, 0};
    unsigned long long* topk_keys = nullptr; size_t topk_keys_n = 0;       // every row's key of up to 8 queries (mocha_match_topk)
    // fp32 banks of at least SCAN16_MIN rows also keep their centred bf16 copy and the rounding residual's norm per row: the
    // few-query matcher scans the copy (half the bytes) and re-ranks exactly what the rounding cannot exclude (match_stream.hip)
    bool scan16 = true;                // mocha_set_option("scan16", 0) scans the fp32 rows themselves
    void* bank16f = nullptr; size_t bank16f_cap = 0; float* bank_rho = nullptr; size_t bank_rho_cap = 0; bool bank16f_valid = false;
    unsigned long long* scan_keys[MAX_SETS] = {nullptr, nullptr, nullptr}; size_t scan_keys_n[MAX_SETS] = {0, 0, 0};
    // option "scan8" (default 0): the centred rows of an fp32 bank as biased bytes + per-row scale + residual bound - an adaptive 1 B / value first
    // stage of the few-query scan (match_scan8.hip); + N x 23 040 B
    bool match_fold = false;           // option "match_fold": the bf16 coarse pass's last K-slab workgroup per tile sums the slabs; the selection reads one (round 6 experiment)
    unsigned* fold_tickets[MAX_SETS] = {nullptr, nullptr, nullptr}; size_t fold_tickets_n[MAX_SETS] = {0, 0, 0};
    bool scan8 = false;
    void* bank8 = nullptr; size_t bank8_cap = 0; float* bank8_scale = nullptr; float* bank8_rho = nullptr; bool bank8_valid = false;
    // many-query matching, round 4 (match_select2.hip): the producer of the centred queries hands over the row statistics of the selection's
    // error bound (match_qstat: ||q - c||^2 and the planes' residual per query), the bf16 coarse pass takes two query planes
    // Used for fp32 banks (demo pair 585 x 585: centre + select 44 us against 68; 128 x 4096: 23 against 30).  bf16 banks stay on
    // mocha_center_bf16 + one plane + mocha_match_select: with ONE query plane the bound admits ~8 rows for the unluckiest query and the
    // LDS-staged kernel re-evaluates them cheaper (128 x 4096: 81 us in all against 103); with TWO planes the selection drops to 15 us but the
    // coarse pass, whose ring then holds a third less of the bank in flight, rises from 39 to 60 us (85 in all; 1024 x 4096: 493 against 360) -
    // mocha_set_option("match_planes", 2) selects that variant (profiles/r04/c_select_ab.txt)
    bool select2 = true;
    int match_planes = 1;              // bf16 query planes of the many-query coarse pass against a bf16 bank
    // round 5: the one-plane coarse pass as mocha_match_pass256 (match_pass.hip: 256-row tiles, bank operands straight into registers) for
    // up to match_pass_max_q queries; 0 = round 4's mocha_match_gemm_bf16_dma.  match_pass_variant: launch_match_pass256's variant bits.
    int match_pass = 0, match_pass_variant = 0, match_pass_max_q = 256;
    // match_pass = 2: mocha_match_pass256 reading the bank from its operand-order image (every wave-level load 1 KB contiguous), built at
    // the first such match (+ 2 B per bank value); match_nt: round 4's kernel with non-temporal bank loads
    void* bank_tile32 = nullptr; size_t bank_tile32_cap = 0; bool bank_tile32_valid = false;
    int match_nt = 1;                  // applied to launches with ONE query tile (Q <= 128): the bank is then read exactly once per launch
    DevBuf match_qstat[MAX_SETS];

    // captured per-window step (mocha_step_graph): one executable graph, re-captured when its key changes
    struct StepGraph {
        hipGraphExec_t exec = nullptr; hipGraph_t graph = nullptr;
        const void *x = nullptr, *mean = nullptr, *sd = nullptr; void *y = nullptr, *idx = nullptr;
        int64_t generation = -1; bool raw = false;
    } step[MAX_SETS];
    hipStream_t cap_stream = nullptr;                 // capture happens on this internal stream (the caller's may be the null stream)
    ncclComm_t comm = nullptr; int comm_rank = 0, comm_size = 1;      // mocha_comm_init
    long long* bcast_hdr = nullptr;                                     // device: {entries, bf16?} header of mocha_bank_broadcast

    // per-launch HIP-event profiling (mocha_profile_start/stop); off in normal operation
    struct ProfRec { std::string kernel, site; hipEvent_t e0, e1; double flops, bytes; };
    bool prof_on = false;
    std::vector<ProfRec> prof;
};

namespace {

int fail(mocha_ctx* c, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
    if (c) c->err = buf; else g_create_error = buf;
    return code;
}

#define HIPCHK(c, expr)                                                                          \
    do {                                                                                         \
        hipError_t e__ = (expr);                                                                 \
        if (e__ != hipSuccess)                                                                   \
            return fail((c), MOCHA_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), __FILE__, __LINE__); \
    } while (0)

int dev_alloc(mocha_ctx* c, float** out, size_t nfloats) {
    void* p = nullptr;
    HIPCHK(c, hipMalloc(&p, std::max<size_t>(nfloats, 4) * sizeof(float)));
    *out = (float*)p;
    c->owned.push_back((float*)p);
    return 0;
}

void dev_free(mocha_ctx* c, float* p) {
    if (!p) return;
    (void)hipFree(p);
    c->owned.erase(std::remove(c->owned.begin(), c->owned.end(), p), c->owned.end());
}

// host -> device copy of one named tensor; a re-load (second load_state_dict on the same context) reuses the
// existing allocation when the size is unchanged and frees it otherwise, so repeated loads do not leak
int upload_to(mocha_ctx* c, std::map<std::string, float*>& tab, std::map<std::string, size_t>& sizes, const std::string& name,
              const std::vector<float>& v) {
    float* d = nullptr;
    auto it = tab.find(name);
    if (it != tab.end() && sizes[name] == v.size()) d = it->second;
    else {
        if (it != tab.end()) { HIPCHK(c, hipDeviceSynchronize()); dev_free(c, it->second); tab.erase(it); }
        int rc = dev_alloc(c, &d, v.size());
        if (rc) return rc;
    }
    HIPCHK(c, hipMemcpy(d, v.data(), v.size() * sizeof(float), hipMemcpyHostToDevice));
    tab[name] = d; sizes[name] = v.size();
    return 0;
}
int upload(mocha_ctx* c, const std::string& name, const std::vector<float>& v) { return upload_to(c, c->w, c->wsize, name, v); }

void build_expectations(mocha_ctx* c) {
    const mocha_cfg& g = c->cfg;
    const int64_t d = g.dim, d4 = g.dim / g.patch, cin = g.C_in, V = g.V;
    auto& e = c->expect;
    e["pos_emb"] = {1, (int64_t)c->ntok, d};
    e["mot_embedding.1.weight"] = {d4, cin, 1, 1};
    e["mot_embedding.1.bias"] = {d4};
    e["mot_embedding.2.blk.gcn.conv.weight"] = {3 * d, d4, 1, 1};
    e["mot_embedding.2.blk.gcn.conv.bias"] = {3 * d};
    e["mot_embedding.2.blk.tcn.weight"] = {d, d, 5, 1};
    e["mot_embedding.2.blk.tcn.bias"] = {d};
    e["mot_embedding.5.blk.gcn.conv.weight"] = {2 * d, d, 1, 1};
    e["mot_embedding.5.blk.gcn.conv.bias"] = {2 * d};
    e["mot_embedding.5.blk.tcn.weight"] = {d, d, 3, 1};
    e["mot_embedding.5.blk.tcn.bias"] = {d};
    for (int dec = 0; dec < 2; ++dec) {
        const char* nm = dec ? "decoder" : "encoder";
        const int depth = dec ? g.dec_depth : g.enc_depth;
        const int64_t inner = dec ? (int64_t)g.dec_heads * g.dec_dim_head : (int64_t)g.enc_heads * g.enc_dim_head;
        const int64_t mlp = dec ? g.dec_mlp : g.enc_mlp;
        for (int l = 0; l < depth; ++l) {
            std::string p = std::string(nm) + ".layers." + std::to_string(l);
            if (dec) {
                e[p + ".0.style.2.weight"] = {2 * d, d};
                e[p + ".0.style.2.bias"] = {2 * d};
                e[p + ".0.style.4.weight"] = {2 * d, 2 * d};
                e[p + ".0.style.4.bias"] = {2 * d};
            }
            e[p + ".1.to_q.1.weight"] = {inner, d};
            e[p + ".1.to_k.1.weight"] = {inner, d};
            e[p + ".1.to_v.weight"] = {inner, d};
            e[p + ".1.to_out.0.weight"] = {d, inner};
            e[p + ".1.to_out.0.bias"] = {d};
            e[p + ".2.net.0.weight"] = {mlp, d};
            e[p + ".2.net.0.bias"] = {mlp};
            e[p + ".2.net.3.weight"] = {d, mlp};
            e[p + ".2.net.3.bias"] = {d};
        }
    }
    e["to_mot.1.blk.gcn.conv.weight"] = {2 * d, d, 1, 1};
    e["to_mot.1.blk.gcn.conv.bias"] = {2 * d};
    e["to_mot.1.blk.tcn.weight"] = {d, d, 3, 1};
    e["to_mot.1.blk.tcn.bias"] = {d};
    e["to_mot.4.blk.gcn.conv.weight"] = {3 * d4, d, 1, 1};
    e["to_mot.4.blk.gcn.conv.bias"] = {3 * d4};
    e["to_mot.4.blk.tcn.weight"] = {d4, d4, 5, 1};
    e["to_mot.4.blk.tcn.bias"] = {d4};
    e["to_mot.6.weight"] = {cin, d4, 1, 1};
    e["to_mot.6.bias"] = {cin};
    (void)V;
}

// registered buffers of the reference state_dict: accepted and cross-checked, never used
bool is_graph_buffer(const std::string& n) {
    return n == "mot_embedding.2.A_j" || n == "to_mot.4.A_j" || n == "mot_embedding.5.A_b" || n == "to_mot.1.A_b" ||
           n == "mot_embedding.3.weight" || n == "to_mot.3.weight";
}

int check_graph_buffer(mocha_ctx* c, const std::string& n, const float* host, const int64_t* shape, int ndim) {
    const Skeleton& sk = c->sk;
    const std::vector<double>* ref = nullptr;
    std::vector<int64_t> shp;
    if (n.size() > 3 && n.compare(n.size() - 3, 3, "A_j") == 0) { ref = &sk.A_j; shp = {3, sk.V, sk.V}; }
    else if (n.size() > 3 && n.compare(n.size() - 3, 3, "A_b") == 0) { ref = &sk.A_b; shp = {2, 6, 6}; }
    else if (n == "mot_embedding.3.weight") { ref = &sk.pool; shp = {sk.V, 6}; }
    else { ref = &sk.unpool; shp = {6, sk.V}; }
    if ((int)shp.size() != ndim) return fail(c, MOCHA_ERR_WEIGHT, "%s: rank %d, expected %zu", n.c_str(), ndim, shp.size());
    for (int i = 0; i < ndim; ++i)
        if (shape[i] != shp[i]) return fail(c, MOCHA_ERR_WEIGHT, "%s: dim %d is %lld, expected %lld", n.c_str(), i, (long long)shape[i], (long long)shp[i]);
    for (size_t i = 0; i < ref->size(); ++i)
        if (std::fabs((double)host[i] - (*ref)[i]) > 1e-6)
            return fail(c, MOCHA_ERR_WEIGHT, "%s: graph constant differs from the regenerated one at %zu (%g vs %g)", n.c_str(), i, host[i], (*ref)[i]);
    return 0;
}

const std::vector<float>& W(mocha_ctx* c, const std::string& n) { return c->host_w.at(n).data; }

// [co][tap*Cin + ci] <- conv weight (Cout, Cin, taps, 1)
std::vector<float> repack_tcn(const std::vector<float>& w, int cout, int cin, int taps) {
    std::vector<float> o((size_t)cout * taps * cin);
    for (int co = 0; co < cout; ++co)
        for (int ci = 0; ci < cin; ++ci)
            for (int t = 0; t < taps; ++t) o[((size_t)co * taps + t) * cin + ci] = w[((size_t)co * cin + ci) * taps + t];
    return o;
}
// adjacency-first form of SpatialConv (net/blocks.py:57-66): [co][k*Cin + ci] <- conv weight (K*Cout, Cin, 1, 1)
std::vector<float> repack_gcn_adjfirst(const std::vector<float>& w, int K, int cout, int cin) {
    std::vector<float> o((size_t)cout * K * cin);
    for (int k = 0; k < K; ++k)
        for (int co = 0; co < cout; ++co)
            for (int ci = 0; ci < cin; ++ci) o[(size_t)co * K * cin + (size_t)k * cin + ci] = w[((size_t)k * cout + co) * cin + ci];
    return o;
}

int ensure_ws(mocha_ctx* c, int B) {
    const int want = std::min(B, c->max_chunk);
    if (c->chunk >= want && c->chunk > 0) return 0;
    if (c->chunk > 0) HIPCHK(c, hipDeviceSynchronize());      // growing: earlier work may still read the old buffers
    // per-window float counts
    const int V = c->cfg.V;
    const size_t T = 90 * 256;
    const std::pair<const char*, size_t> plan[] = {
        {"hbar", 360 * 192}, {"ybar", 360 * 256}, {"u", 90 * 1280}, {"x5", T}, {"xA", 90 * 512}, {"t1", T}, {"xa", T}, {"xb", T},
        {"qkv", 90 * 3072}, {"ao", 90 * 1024}, {"hff", (size_t)90 * std::max(512, std::max(c->cfg.enc_mlp, c->cfg.dec_mlp))}, {"kin", T}, {"xad", T}, {"qin", T},
        {"smean", 256}, {"s1", 512 * 8}, {"gb", 512 * 8}, {"smean64", 512}, {"s1d", 512 * 8 * 2}, {"qc", T}, {"g", 90 * 192}, {"y2c", (size_t)15 * V * 64},
        {"z", (size_t)60 * V * 64}, {"enc_s", T}, {"enc_c", T}, {"qnm", T}, {"sel", T}, {"dec", T},
        // the decoder attention's pre-split key / value images (attention_kv.hip): 295 KB per window, planned only while option
        // "attention_kv" is on (setting it re-plans the workspaces)
        {"kvimg", c->attn_kv ? (size_t)ATTN_KV_IMG_BYTES / 4 : (size_t)4},
    };
    // free old workspaces
    for (int set = 0; set < mocha_ctx::MAX_SETS; ++set) {
        for (auto& kv : c->wss[set]) {
            dev_free(c, kv.second.p);
        }
        c->wss[set].clear();
        if (c->idx_ws[set]) { (void)hipFree(c->idx_ws[set]); c->idx_ws[set] = nullptr; }
    }
    const bool dual_sets = c->dual_stream && want >= c->dual_min / 2;
    const int nsets = std::max(dual_sets ? 2 : 1, c->lanes);
    for (int set = 0; set < nsets; ++set) {
        // a lane's set serves per-window steps: a handful of windows; dual_stream's second set takes half of every large batch
        const int wset = (set == 0 || (set == 1 && dual_sets)) ? want : std::min(want, 8);
        for (auto& pl : plan) {
            DevBuf b; b.n = pl.second * (size_t)wset;
            int rc = dev_alloc(c, &b.p, b.n);
            if (rc) return rc;
            c->wss[set][pl.first] = b;
        }
        void* ip = nullptr;
        HIPCHK(c, hipMalloc(&ip, sizeof(int32_t) * (size_t)wset));
        c->idx_ws[set] = (int32_t*)ip;
    }
    c->idx_ws_n = want;
    c->chunk = want;
    c->dual_ready = dual_sets;
    c->generation++;
    return 0;
}

float* WS(mocha_ctx* c, const char* n) { return c->wss[c->cur].at(n).p; }
InormExtra IEX(const mocha_ctx* c) { InormExtra e; e.split_max = c->inorm_split_max; return e; }     // the instance norm's extras with this context's launch tuning
float* DW(mocha_ctx* c, const std::string& n) { return c->w.at(n); }

// Every kernel launch goes through LAUNCH: error check, and when profiling is on a HIP event
// pair on the launch stream around it, tagged with the kernel symbol and the call site.
int prof_begin(mocha_ctx* c, hipStream_t s, const char* kernel, const char* site, double flops, double bytes) {
    mocha_ctx::ProfRec r{kernel, site, nullptr, nullptr, flops, bytes};
    HIPCHK(c, hipEventCreate(&r.e0));
    HIPCHK(c, hipEventCreate(&r.e1));
    HIPCHK(c, hipEventRecord(r.e0, s));
    c->prof.push_back(r);
    return 0;
}

#define LAUNCH(c, s, kernel, site, flops, bytes, expr)                                                       \
    do {                                                                                                     \
        if ((c)->prof_on) { int rc__ = prof_begin((c), (s), (kernel), (site), (flops), (bytes)); if (rc__) return rc__; } \
        hipError_t e__ = (expr);                                                                             \
        if (e__ != hipSuccess)                                                                               \
            return fail((c), MOCHA_ERR_HIP, "launch %s: %s (%s:%d)", #expr, hipGetErrorString(e__), __FILE__, __LINE__); \
        if ((c)->prof_on) HIPCHK((c), hipEventRecord((c)->prof.back().e1, (s)));                             \
    } while (0)

const char* gemm_kernel_name(const GemmParams& p) {
    if (gemm_is_skinny16(p)) return "mocha_gemm_skinny16";
    if (gemm_is_skinny(p)) return "mocha_gemm_skinny";
    if (gemm_is_small(p)) return "mocha_gemm_f32<64,2,2,1,1>";
    return gemm_is_narrow(p) ? "mocha_gemm_f32<64,4,1,1,2>" : "mocha_gemm_f32<128,2,2,2,2>";
}

// packed weight image of the bf16x3 engine for (W, N, K); made on first use (not while the stream is capturing: the caller
// then takes the exact-f32 kernel), dropped whenever weights are (re-)finalised
int x3_image(mocha_ctx* c, hipStream_t s, const GemmParams& p, const unsigned short** out) {
    *out = nullptr;
    const auto key = std::make_tuple(p.W, p.N, p.K);
    auto it = c->x3w.find(key);
    if (it != c->x3w.end()) { *out = it->second; return 0; }
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &st) != hipSuccess || st != hipStreamCaptureStatusNone) return 0;
    float* d = nullptr;
    int rc = dev_alloc(c, &d, (gemm_x3_packed_elems(p.N, p.K) + 1) / 2);
    if (rc) return rc;
    HIPCHK(c, launch_pack_x3(p.W, p.N, p.K, reinterpret_cast<unsigned short*>(d), s));
    HIPCHK(c, hipStreamSynchronize(s));             // once per weight: the image may be read from the other stream of a dual-stream step
    c->x3w[key] = reinterpret_cast<unsigned short*>(d);
    *out = c->x3w[key];
    return 0;
}

// A graph captured after warm-up has the image pointers baked into its mocha_gemm_x3 nodes: dropping them is a buffer
// replacement like any other, so the generation moves and every holder re-captures.
void x3_drop_images(mocha_ctx* c) {
    if (!c->x3w.empty()) c->generation++;
    for (auto& kv : c->x3w) dev_free(c, reinterpret_cast<float*>(kv.second));
    c->x3w.clear();
}

// packed two-plane fp16 image + per-row inverse scales of (W, N, K) for gemm_h2.hip; made on first use like x3_image
int h2_image(mocha_ctx* c, hipStream_t s, const GemmParams& p, mocha_ctx::H2Img* out) {
    *out = mocha_ctx::H2Img{nullptr, nullptr};
    const auto key = std::make_tuple(p.W, p.N, p.K);
    auto it = c->h2w.find(key);
    if (it != c->h2w.end()) { *out = it->second; return 0; }
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &st) != hipSuccess || st != hipStreamCaptureStatusNone) return 0;
    float *d = nullptr, *wi = nullptr;
    int rc = dev_alloc(c, &d, (gemm_h2_packed_elems(p.N, p.K) + 1) / 2);
    if (!rc) rc = dev_alloc(c, &wi, (size_t)p.N);
    if (rc) return rc;
    HIPCHK(c, launch_pack_h2(p.W, p.N, p.K, reinterpret_cast<unsigned short*>(d), wi, s));
    HIPCHK(c, hipStreamSynchronize(s));
    c->h2w[key] = mocha_ctx::H2Img{reinterpret_cast<unsigned short*>(d), wi};
    c->generation++;            // a step graph captured before this image existed took the three-plane kernels for this weight: its holder re-captures
    *out = c->h2w[key];
    return 0;
}
void h2_drop_images(mocha_ctx* c) {
    if (!c->h2w.empty()) c->generation++;
    for (auto& kv : c->h2w) { dev_free(c, reinterpret_cast<float*>(kv.second.img)); dev_free(c, kv.second.w_inv); }
    c->h2w.clear();
}
// activation bounds of the two-plane fp16 engine (see mocha_ctx::amax): a stage's slots are zeroed when the stage starts
int amax_alloc(mocha_ctx* c) {                                       // once, when the option is switched on (never inside a capture)
    if (c->amax) return 0;
    const size_t n = (size_t)mocha_ctx::MAX_SETS * mocha_ctx::AMAX_STAGES * mocha_ctx::AMAX_SLOTS;
    int rc = dev_alloc(c, &c->amax, (n + 1) * mocha_ctx::AMAX_WIN);
    if (rc) return rc;
    c->amax_in = c->amax + n * mocha_ctx::AMAX_WIN;
    HIPCHK(c, hipMemset(c->amax, 0, n * mocha_ctx::AMAX_WIN * sizeof(float)));
    const std::vector<float> k(mocha_ctx::AMAX_WIN, 9.5f);
    HIPCHK(c, hipMemcpy(c->amax_in, k.data(), k.size() * sizeof(float), hipMemcpyHostToDevice));
    return 0;
}
// b: windows of this chunk - up to 4 every GEMM of the path runs on the few-rows kernels (gemm_is_skinny), which neither read nor leave
// bounds: the stage then hands out no slots (no memset / absmax launches in the streamed per-window step)
int amax_begin(mocha_ctx* c, int stage, hipStream_t s, int b) {
    c->amax_idle = !c->gemm_h2 || !c->amax || b < 5 || b > mocha_ctx::AMAX_WIN;
    if (c->amax_idle) return 0;
    c->amax_stage = stage; c->amax_next = 0;
    const size_t base = ((size_t)c->cur * mocha_ctx::AMAX_STAGES + stage) * mocha_ctx::AMAX_SLOTS;
    std::fill(c->amax_ok.begin() + base, c->amax_ok.begin() + base + mocha_ctx::AMAX_SLOTS, 0);
    // the first b entries of every slot the stage can take: one strided memset
    HIPCHK(c, hipMemset2DAsync(c->amax + base * mocha_ctx::AMAX_WIN, mocha_ctx::AMAX_WIN * sizeof(float), 0, (size_t)b * sizeof(float),
                               (size_t)mocha_ctx::AMAX_USED[stage], s));
    return 0;
}
float* amax_slot(mocha_ctx* c) {                                      // a fresh (zero) bound vector of the current stage, or null when the engine is off
    if (c->amax_idle || !c->gemm_h2 || !c->amax || c->amax_next >= mocha_ctx::AMAX_USED[c->amax_stage]) return nullptr;
    return c->amax + (((size_t)c->cur * mocha_ctx::AMAX_STAGES + c->amax_stage) * mocha_ctx::AMAX_SLOTS + c->amax_next++) * mocha_ctx::AMAX_WIN;
}
ptrdiff_t amax_index(const mocha_ctx* c, const float* slot) {         // slot number inside the arena, or -1 (amax_in / foreign)
    if (!c->amax || !slot || slot < c->amax) return -1;
    const ptrdiff_t i = (slot - c->amax) / mocha_ctx::AMAX_WIN;
    return i < (ptrdiff_t)c->amax_ok.size() ? i : -1;
}
const float* amax_use(const mocha_ctx* c, const float* slot) {        // the bound if a launch wrote it (or it lies outside the arena: the caller checked), else null
    if (!c->gemm_h2 || !slot || !c->amax) return nullptr;
    const ptrdiff_t i = amax_index(c, slot);
    if (i >= 0) return c->amax_ok[i] ? slot : nullptr;
    return slot;
}
// *slot[w] = mul * max |window w of x| + add for the chunk's windows of `per` floats each (mocha_absmax)
int amax_measure(mocha_ctx* c, hipStream_t s, const float* x, int nwin, long long per, float** slot, float mul = 1.f, float add = 0.f) {
    *slot = amax_slot(c);
    if (!*slot) return 0;
    LAUNCH(c, s, "mocha_absmax", "h2.absmax", 0.0, 4.0 * nwin * per, launch_absmax(x, nwin, per, *slot, s, mul, add));
    c->amax_ok[amax_index(c, *slot)] = 1;
    return 0;
}

int gemm(mocha_ctx* c, hipStream_t s, const char* site, const GemmParams& p0) {
    GemmParams p = p0; p.tile64_below = c->gemm_tile64_below; p.persistent = c->gemm_persistent; p.persistent_max_n = c->gemm_persistent_max_n;
    if (p.rows_per_win <= 0) p.rows_per_win = 90;      // rows of a launch per window: 90 tokens unless the site says otherwise (to_mot's joint rows)
    if (c->gemm_h2 && p.a_amax && gemm_h2_supports(p)) {
        mocha_ctx::H2Img img;
        int rc = h2_image(c, s, p, &img);
        if (rc) return rc;
        if (img.img) {
            GemmParams q = p; q.Wh2 = img.img; q.w_inv = img.w_inv;
            const double fl = 2.0 * p.M * (double)p.N * p.K;
            const double by = 4.0 * ((double)p.M * p.K / (p.gather ? p.ntaps : 1) * (p.R) + (double)p.N * p.K + (double)p.M * p.N + (p.residual ? (double)p.M * p.N : 0.0));
            LAUNCH(c, s, "mocha_gemm_h2", site, fl, by, launch_gemm_h2(q, s));
            { const ptrdiff_t i = amax_index(c, p.c_amax); if (i >= 0) c->amax_ok[i] = 1; }
            return 0;
        }
    }
    const double flops = 2.0 * p.M * (double)p.N * p.K;
    // algorithmic bytes: every operand once - activations, weights, the output, and the residual matrix where the epilogue adds one
    const double bytes = 4.0 * ((double)p.M * p.K / (p.gather ? p.ntaps : 1) * (p.R) + (double)p.N * p.K + (double)p.M * p.N * p.ksplit +
                                (p.residual ? (double)p.M * p.N : 0.0));
    if (c->gemm_x3 && gemm_x3_supports(p)) {
        const unsigned short* img = nullptr;
        int rc = x3_image(c, s, p, &img);
        if (rc) return rc;
        if (img) {
            GemmParams q = p; q.Wsplit = img;
            if (c->gemm_x3r_min_n > 0 && p.N >= c->gemm_x3r_min_n && gemm_x3r_supports(q)) {      // same bits, another schedule (gemm_x3r.hip)
                LAUNCH(c, s, "mocha_gemm_x3r", site, flops, bytes, launch_gemm_x3r(q, s));
                return 0;
            }
            LAUNCH(c, s, "mocha_gemm_x3", site, flops, bytes, launch_gemm_x3(q, s));
            return 0;
        }
    }
    LAUNCH(c, s, gemm_kernel_name(p), site, flops, bytes, launch_gemm(p, s));
    return 0;
}
// the Generator's attention (90 tokens, head dim 128 / 256): plane products on the bf16 pipe unless switched off
const char* attn_kernel_name(const mocha_ctx* c, int DH, long long pairs = 1 << 30) {
    const bool x3 = c->attn_x3 && (DH == 128 || DH == 256);
    if (x3 && DH == 256 && pairs <= c->attn_split_max) return "mocha_attention_x3_split<256>";
    return x3 ? (DH == 128 ? "mocha_attention_x3<128>" : "mocha_attention_x3<256>")
              : (DH == 64 ? "mocha_attention_f32<64>" : DH == 128 ? "mocha_attention_f32<128>" : "mocha_attention_f32<256>");
}
hipError_t attention(const mocha_ctx* c, const AttnParams& a0, hipStream_t s) {
    AttnParams a = a0; a.split_max = c->attn_split_max;
    return (c->attn_x3 && (a.dh == 128 || a.dh == 256) && a.nq <= 96 && a.nk <= 96) ? launch_attention_x3(a, s) : launch_attention(a, s);
}
#define GEMM(c, s, site, p) do { int rc__ = gemm((c), (s), (site), (p)); if (rc__) return rc__; } while (0)

GemmParams plain(const float* A, int lda, const float* Wt, float* C, int ldc, int M, int N, int K) {
    GemmParams p;
    p.A = A; p.lda = lda; p.W = Wt; p.C = C; p.ldc = ldc; p.M = M; p.N = N; p.K = K;
    return p;
}

// ---------------------------------------------------------------- stage pipelines on one chunk
// mot_embedding (model.py:42-50) for b windows: X -> tokens (b*90, 256)
// X2 / b2: an optional second clip whose windows follow the first one's in every workspace (pair step: both clips of a
// demo pair share the launches from the gcn conv on, so each launch has twice the tiles)
int run_embed(mocha_ctx* c, const float* X, int b, float* tokens, bool add_pos, hipStream_t s, bool raw = false,
              const float* X2 = nullptr, int b2 = 0) {
    const int V = c->cfg.V;
    const int nn = (V + 1) * c->cfg.C_in;
    if (raw && !c->pose_norm) return fail(c, MOCHA_ERR_STATE, "raw input needs mocha_set_pose_norm first");
    // two-plane fp16 engine (z-scored input, fused path): |u| <= |conv1(x)| <= emb_l1 max|x| + emb_bmax (LeakyReLU, the column-normalised
    // adjacency x pool mix and the 4-frame means do not raise a magnitude) - one pass over the poses gives the first launch's bound,
    // every GEMM's epilogue the next one's; the tokens' bound goes on to the encoder
    { int rc = amax_begin(c, 3, s, b + b2); if (rc) return rc; }
    float* u_amax = nullptr;
    c->amax_tok_of[c->cur] = nullptr;
    if (c->gemm_h2 && !c->amax_idle && !raw && c->fold_joint && c->gemm_x3 && c->embed_sums && c->emb_l1 > 0.f && (u_amax = amax_slot(c))) {
        const long long per = 60ll * V * c->cfg.C_in;
        LAUNCH(c, s, "mocha_absmax", "h2.absmax", 0.0, 4.0 * b * per, launch_absmax(X, b, per, u_amax, s, c->emb_l1, c->emb_bmax));
        if (X2 && b2 > 0) LAUNCH(c, s, "mocha_absmax", "h2.absmax", 0.0, 4.0 * b2 * per, launch_absmax(X2, b2, per, u_amax + b, s, c->emb_l1, c->emb_bmax));
        c->amax_ok[amax_index(c, u_amax)] = 1;
    }
    float* x5_amax = nullptr;
    if (c->fold_joint && c->gemm_x3 && c->embed_sums) {
        // conv1 + lrelu + adjacency + joint->part pool, and the 4-frame sums of the five taps, in one kernel: the frame rows stay in LDS
        LAUNCH(c, s, "mocha_embed_sums_x3", "emb.front_sums", b * 60.0 * V * 64 * (2.0 * 15 + 2.0 * 18) + b * 90.0 * 960 * 4,
               b * 4.0 * (60.0 * V * 15 + 90.0 * 960),
               launch_embed_sums(X, DW(c, "emb.W1"), DW(c, "emb.b1"), DW(c, "AP"), WS(c, "u"), b, V, c->cfg.C_in,
                                 raw ? c->pose_norm : nullptr, raw ? c->pose_norm + nn : nullptr, raw ? 1 : 0, s, c->embed_max_wgs));
        if (X2 && b2 > 0) {
            LAUNCH(c, s, "mocha_embed_sums_x3", "emb.front_sums", b2 * 60.0 * V * 64 * (2.0 * 15 + 2.0 * 18) + b2 * 90.0 * 960 * 4,
                   b2 * 4.0 * (60.0 * V * 15 + 90.0 * 960),
                   launch_embed_sums(X2, DW(c, "emb.W1"), DW(c, "emb.b1"), DW(c, "AP"), WS(c, "u") + (size_t)b * 90 * 960, b2, V, c->cfg.C_in,
                                     raw ? c->pose_norm : nullptr, raw ? c->pose_norm + nn : nullptr, raw ? 1 : 0, s, c->embed_max_wgs));
            b += b2;
        }
        GemmParams gf = plain(WS(c, "u"), 960, DW(c, "emb.Wc"), WS(c, "x5"), 256, b * 90, 256, 960);
        gf.rowbias = DW(c, "emb.rbc"); gf.rb_mod = 6;
        gf.a_amax = amax_use(c, u_amax); gf.c_amax = x5_amax = amax_slot(c);
        GEMM(c, s, "emb.joint_block", gf);
    } else {
    // conv1 + lrelu + adjacency + joint->part pool (commuted)                  model.py:44-46
    const char* efk = c->gemm_x3 ? "mocha_embed_front_x3" : "mocha_embed_front";
    LAUNCH(c, s, efk, "emb.front", b * 60.0 * V * 64 * (2.0 * 15 + 2.0 * 18), b * 60.0 * (V * 15 + 6 * 192) * 4,
           launch_embed_front(X, DW(c, "emb.W1"), DW(c, "emb.b1"), DW(c, "AP"), WS(c, "hbar"), b * 60, V, c->cfg.C_in,
                              raw ? c->pose_norm : nullptr, raw ? c->pose_norm + nn : nullptr, raw ? 1 : 0, s, c->gemm_x3, c->embed_max_wgs));
    if (X2 && b2 > 0) {
        LAUNCH(c, s, efk, "emb.front", b2 * 60.0 * V * 64 * (2.0 * 15 + 2.0 * 18), b2 * 60.0 * (V * 15 + 6 * 192) * 4,
               launch_embed_front(X2, DW(c, "emb.W1"), DW(c, "emb.b1"), DW(c, "AP"), WS(c, "hbar") + (size_t)b * 360 * 192, b2 * 60, V,
                                  c->cfg.C_in, raw ? c->pose_norm : nullptr, raw ? c->pose_norm + nn : nullptr, raw ? 1 : 0, s, c->gemm_x3, c->embed_max_wgs));
        b += b2;
    }
    if (c->fold_joint) {
        // gcn conv folded into the temporal conv (mocha_finalize_weights: emb.Wc): the 4-frame sums of the five taps are taken on
        // the 192 adjacency-mixed channels and ONE GEMM, K = 5 x 192, does both convolutions and the AvgPool
        LAUNCH(c, s, "mocha_window_sums", "emb.window_sums", b * 90.0 * 960 * 4, b * 4.0 * (360.0 * 192 + 90.0 * 960),
               launch_window_sums(WS(c, "hbar"), WS(c, "u"), b * 90, 192, s));
        GemmParams gf = plain(WS(c, "u"), 960, DW(c, "emb.Wc"), WS(c, "x5"), 256, b * 90, 256, 960);
        gf.rowbias = DW(c, "emb.rbc"); gf.rb_mod = 6;
        GEMM(c, s, "emb.joint_block", gf);
    } else {
    // gcn 1x1 conv on the pooled operand: (b*360, 192) x (256,192)^T + pooled bias
    GemmParams g1 = plain(WS(c, "hbar"), 192, DW(c, "emb.Wg"), WS(c, "ybar"), 256, b * 360, 256, 192);
    g1.rowbias = DW(c, "emb.rbg"); g1.rb_mod = 6;
    GEMM(c, s, "emb.gcn_joint", g1);
    // temporal conv k=5 (reflect) fused with AvgPool2d((4,1)):  (b*90, 5*256) x (256, 1280)^T   blocks.py:112-118, model.py:47
    GemmParams g2 = plain(WS(c, "u"), 1280, DW(c, "emb.Wt"), WS(c, "x5"), 256, b * 90, 256, 1280);
    g2.bias = DW(c, "emb.bt");
    {
        LAUNCH(c, s, "mocha_window_sums", "emb.window_sums", b * 90.0 * 1280 * 4, b * 4.0 * (360.0 * 256 + 90.0 * 1280),
               launch_window_sums(WS(c, "ybar"), WS(c, "u"), b * 90, 256, s));
    }
    GEMM(c, s, "emb.tcn_joint_pool", g2);
    }
    }
    // body block                                                               model.py:48,137-162
    LAUNCH(c, s, "mocha_body_front", "emb.body_front", b * 90.0 * 512 * 12, b * 90.0 * (256 + 512) * 4, launch_body_front(WS(c, "x5"), DW(c, "A_b"), WS(c, "xA"), b * 15, s));
    GemmParams g3 = plain(WS(c, "xA"), 512, DW(c, "emb.Wgb"), WS(c, "t1"), 256, b * 90, 256, 512);
    g3.rowbias = DW(c, "emb.rbb"); g3.rb_mod = 6;
    g3.a_amax = amax_use(c, x5_amax); g3.c_amax = amax_slot(c);        // body_front: LeakyReLU + a column-normalised mix of x5
    GEMM(c, s, "emb.gcn_body", g3);
    GemmParams g4 = plain(WS(c, "t1"), 256, DW(c, "emb.Wtb"), tokens, 256, b * 90, 256, 768);
    g4.gather = 1; g4.T_out = 15; g4.V = 6; g4.ntaps = 3; g4.pad = 1; g4.stride = 1; g4.R = 1; g4.T_full = 15;
    g4.tshift = 0; g4.Cc = 256; g4.T_src = 15; g4.bias = DW(c, "emb.btb");
    if (add_pos) { g4.rowbias = DW(c, "pos_emb"); g4.rb_mod = 90; }           // model.py:88
    g4.a_amax = amax_use(c, g3.c_amax); g4.c_amax = amax_slot(c);
    GEMM(c, s, "emb.tcn_body", g4);
    if (amax_use(c, g4.c_amax)) { c->amax_tok_of[c->cur] = tokens; c->amax_tok[c->cur] = g4.c_amax; }
    return 0;
}

// one transformer layer's attention output projection + FF (net/transformer.py:91-94), shared by enc/dec
// ao_amax: per-window bounds on |ao| (device vector, two-plane fp16 engine) or null; *out_amax: the slot the last GEMM leaves the output's per-window maxima in
int run_out_ff(mocha_ctx* c, const std::string& p, const float* ao, int inner, const float* resid, int M, int mlp,
               float* out, hipStream_t s, const char* wo = ".Wo", const float* ao_amax = nullptr, float** out_amax = nullptr) {
    GemmParams o = plain(ao, inner, DW(c, p + wo), WS(c, "xb"), 256, M, 256, inner);
    o.bias = DW(c, p + ".bo"); o.residual = resid; o.ldr = 256;
    o.a_amax = amax_use(c, ao_amax); o.c_amax = amax_slot(c);
    GEMM(c, s, "xf.out_proj", o);
    GemmParams f1 = plain(WS(c, "xb"), 256, DW(c, p + ".W1"), WS(c, "hff"), mlp, M, mlp, 256);
    f1.bias = DW(c, p + ".b1"); f1.act = 1;
    f1.a_amax = amax_use(c, o.c_amax); f1.c_amax = amax_slot(c);
    GEMM(c, s, "xf.ff1", f1);
    GemmParams f2 = plain(WS(c, "hff"), mlp, DW(c, p + ".W2"), out, 256, M, 256, mlp);
    f2.bias = DW(c, p + ".b2"); f2.residual = WS(c, "xb"); f2.ldr = 256;
    f2.a_amax = amax_use(c, f1.c_amax); f2.c_amax = amax_slot(c);
    GEMM(c, s, "xf.ff2", f2);
    if (out_amax) *out_amax = f2.c_amax;
    return 0;
}

// encoder (model.py:53-59; net/transformer.py:90-95 with adain=False)
int run_encoder(mocha_ctx* c, const float* tokens, int b, float* encoded, hipStream_t s) {
    const int M = b * 90, H = c->cfg.enc_heads, DH = c->cfg.enc_dim_head, inner = H * DH;
    const float* x = tokens;
    // two-plane fp16 engine: every launch scales its activations by a bound on them - the tokens' largest magnitude is measured, every
    // GEMM's epilogue leaves its output's, and the attention's rows are convex combinations of value rows (a slice of qkv)
    int rc0 = amax_begin(c, 0, s, b); if (rc0) return rc0;
    float* x_amax = nullptr;
    if (c->gemm_h2 && !c->amax_idle) {
        if (tokens == c->amax_tok_of[c->cur] && amax_use(c, c->amax_tok[c->cur])) x_amax = c->amax_tok[c->cur];      // the embedding's last GEMM left it
        else { rc0 = amax_measure(c, s, tokens, b, 90ll * 256, &x_amax); if (rc0) return rc0; }
    }
    c->amax_enc_of[c->cur] = nullptr;
    for (int l = 0; l < c->cfg.enc_depth; ++l) {
        const std::string p = "enc" + std::to_string(l);
        GemmParams q = plain(x, 256, DW(c, p + ".Wqkv"), WS(c, "qkv"), 3 * inner, M, 3 * inner, 256);
        q.a_amax = amax_use(c, x_amax); q.c_amax = amax_slot(c);
        GEMM(c, s, "enc.qkv", q);
        AttnParams a{WS(c, "qkv"), WS(c, "qkv") + inner, WS(c, "qkv") + 2 * inner, WS(c, "ao"),
                     3 * inner, 3 * inner, 3 * inner, inner, b, H, DH, 90, 90, (float)std::pow((double)DH, -0.5)};
        LAUNCH(c, s, attn_kernel_name(c, DH), "enc.attn", 4.0 * b * H * 90.0 * 90 * DH,
               4.0 * M * 4 * inner, attention(c, a, s));
        float* out = (l == c->cfg.enc_depth - 1) ? encoded : WS(c, "xa");
        int rc = run_out_ff(c, p, WS(c, "ao"), inner, x, M, c->cfg.enc_mlp, out, s, ".Wo", q.c_amax, &x_amax);
        if (rc) return rc;
        x = out;
    }
    if (amax_use(c, x_amax)) { c->amax_enc_of[c->cur] = encoded; c->amax_enc[c->cur] = x_amax; }
    return 0;
}

// the style MLPs of every decoder layer in float64 (net/transformer.py:102-107): gb (rows, 512 L) = [gamma | beta] per layer, from the
// float64 token means; hidden = rows x 512 L doubles of scratch
int run_style_f64(mocha_ctx* c, const double* mean64, double* hidden, float* gb, int rows, hipStream_t s) {
    const int L = c->cfg.dec_depth;
    LAUNCH(c, s, "mocha_linear_f64", "dec.style1", 2.0 * rows * 512.0 * L * 256, rows * 8.0 * (256 + 512 * L) + 8.0 * 512 * L * 256,
           launch_linear_f64(mean64, 256, 0, c->w64.at("dec.Ws1_all"), c->w64.at("dec.bs1_all"), hidden, nullptr, 512 * L, rows, 512 * L, 256, 1, 2, s));
    LAUNCH(c, s, "mocha_linear_f64", "dec.style2", 2.0 * rows * 512.0 * L * 512, rows * (8.0 + 4.0) * 512 * L + 8.0 * L * 512 * 512,
           launch_linear_f64(hidden, 512 * L, 512, c->w64.at("dec.Ws2_all"), c->w64.at("dec.bs2_all"), nullptr, gb, 512 * L, rows, 512, 512, L, 0, s));
    return 0;
}

// style MLPs of every layer at once for the b windows whose token means the instance norm left in the workspace ("smean" fp32,
// "smean64" float64): Linear 256->512, LeakyReLU, Linear 512->512 per layer -> "gb" (b, 512 L)          net/transformer.py:102-107
int run_style(mocha_ctx* c, int b, hipStream_t s) {
    const int L = c->cfg.dec_depth;
    if (c->style_f64)
        return run_style_f64(c, reinterpret_cast<const double*>(WS(c, "smean64")), reinterpret_cast<double*>(WS(c, "s1d")), WS(c, "gb"), b, s);
    GemmParams s1 = plain(WS(c, "smean"), 256, DW(c, "dec.Ws1_all"), WS(c, "s1"), 512 * L, b, 512 * L, 256);
    s1.bias = DW(c, "dec.bs1_all"); s1.act = 2;
    GEMM(c, s, "dec.style1", s1);
    if (b <= 192 || L == 1) {
        // a handful of windows: launch count matters, the block-diagonal matrix's zero half does not
        GemmParams s2 = plain(WS(c, "s1"), 512 * L, DW(c, "dec.Ws2_blk"), WS(c, "gb"), 512 * L, b, 512 * L, 512 * L);
        s2.bias = DW(c, "dec.bs2_all");
        GEMM(c, s, "dec.style2", s2);
    } else {
        for (int l = 0; l < L; ++l) {                     // large batches: layer l's 512 x 512 on its own slice of the hidden activations
            const std::string p = "dec" + std::to_string(l);
            GemmParams s2 = plain(WS(c, "s1") + (size_t)l * 512, 512 * L, DW(c, p + ".Ws2"), WS(c, "gb") + (size_t)l * 512, 512 * L, b, 512, 512);
            s2.bias = DW(c, p + ".bs2");
            GEMM(c, s, "dec.style2", s2);
        }
    }
    return 0;
}

// can the decoder read a bank entry's constants (IN(cha), gamma / beta) in place through frame_index?  (the folded decoder on the plane
// attention kernel; the image variant and the un-folded projections want contiguous per-window copies)
bool dec_cache_ok(const mocha_ctx* c) {
    return c->bank_dec_cache && c->style_f64 && c->fold_decoder && c->cfg.dec_dim_head == 256 && c->attn_x3 && !c->attn_kv;
}

// What the decoder derives from bank entries alone, for rows [0, N) of `enc`: kin = IN(entry) (the folded decoder's keys) and gb = every
// layer's AdaIN gamma / beta (float64 style MLP of the entry's float64 token mean).  mean64 / hidden: scratch for `cap` rows.
int build_dec_consts(mocha_ctx* c, const float* enc, int64_t N, float* kin, float* gb, double* mean64, double* hidden, int64_t cap, hipStream_t s) {
    const size_t D = 90 * 256;
    const int L = c->cfg.dec_depth;
    for (int64_t r0 = 0; r0 < N; r0 += cap) {
        const int rows = (int)std::min<int64_t>(cap, N - r0);
        InormExtra ex = IEX(c); ex.mean64 = mean64;
        LAUNCH(c, s, "mocha_instnorm", "bank.in_cha", 0.0, rows * 90.0 * 256 * 4 * 2,
               launch_instnorm(enc + (size_t)r0 * D, kin + (size_t)r0 * D, nullptr, nullptr, nullptr, nullptr, rows, 90, s, &ex));
        int rc = run_style_f64(c, mean64, hidden, gb + (size_t)r0 * 512 * L, rows, s);
        if (rc) return rc;
    }
    return 0;
}

// decoder (model.py:62-68; net/transformer.py:90-121 with adain=True)
// gather_table / gather_idx: cha is cha_encoded[frame_index] (test_fullframework.py:298, 465).  With the bank's cached constants
// (kin_table = IN of every entry, gb_table = its gamma / beta: build_dec_consts) the decoder reads everything that depends on the matched
// entry IN PLACE through the indices - no instance norm of cha, no style MLP, no gathered copy; without them the first kernel gathers
// the rows itself, leaves the copy in the "sel" workspace (cha may then be null) and the constants are computed here.
int run_decoder(mocha_ctx* c, const float* src, const float* cha, int b, float* outp, hipStream_t s,
                const float* gather_table = nullptr, const int32_t* gather_idx = nullptr, long long gather_rows = 0,
                const float* kin_table = nullptr, const float* gb_table = nullptr) {
    const int M = b * 90, H = c->cfg.dec_heads, DH = c->cfg.dec_dim_head, inner = H * DH, L = c->cfg.dec_depth;
    const bool cached = gather_table && gather_idx && kin_table && gb_table && dec_cache_ok(c);
    // IN(cha) feeds every layer's keys; mean over tokens of cha feeds every layer's style MLP
    // Folded decoder at batch size: keys IN(cha) and values cha are the same for every head and layer - the instance norm writes them once
    // as pre-split bf16 plane images (InormExtra::kvimg) and every layer's attention reads those (attention_kv.hip); no fp32 copies
    const bool use_kv = c->fold_decoder && DH == 256 && c->attn_x3 && c->attn_kv && H >= 2 && H % 2 == 0 && (long long)b * H > c->attn_split_max;
    // two-plane fp16 engine: the attention's rows are convex combinations of the value rows - the matched bank entries (their largest
    // magnitude is taken at mocha_bank_set) or the character windows this call encoded; the query GEMM's operand is an instance-normalised
    // token (|z| <= (n - 1) / sqrt(n)); without a bound a launch stays on the bf16 planes
    { int rc = amax_begin(c, 1, s, b); if (rc) return rc; }
    const float* v_amax = nullptr;
    if (c->gemm_h2 && !c->amax_idle) {
        const float* vsrc = gather_table ? gather_table : cha;
        // per source window: its matched entry's bound (gathered through the indices), or the same window's character features
        const float* table = nullptr;
        if (vsrc && vsrc == c->bank_enc && c->amax_bank_ok) table = c->amax_bank;
        else if (vsrc && vsrc == c->amax_enc_of[c->cur]) table = amax_use(c, c->amax_enc[c->cur]);
        if (table && gather_table) {
            float* slot = amax_slot(c);
            if (slot) {
                LAUNCH(c, s, "mocha_gather_f32", "h2.gather", 0.0, 12.0 * b, launch_gather_f32(table, gather_idx, gather_rows, slot, b, s));
                c->amax_ok[amax_index(c, slot)] = 1;
                v_amax = slot;
            }
        } else if (table) v_amax = table + (cha - vsrc) / (90 * 256);          // cha = rows of the encoder's output: window for window
        else if (cha && !gather_table) {            // caller-supplied character features (mocha_decoder): measured
            float* slot = nullptr;
            int rc = amax_measure(c, s, cha, b, 90ll * 256, &slot); if (rc) return rc;
            v_amax = amax_use(c, slot);
        }
    }
    float* x_amax = nullptr;
    c->amax_dec_of[c->cur] = nullptr;
    if (!cached) {
        InormExtra ex = IEX(c);
        if (gather_table) { ex.table = gather_table; ex.row_idx = gather_idx; ex.table_rows = gather_rows; ex.copy_out = use_kv ? nullptr : WS(c, "sel"); }
        if (use_kv) ex.kvimg = reinterpret_cast<unsigned short*>(WS(c, "kvimg"));
        if (c->style_f64) ex.mean64 = reinterpret_cast<double*>(WS(c, "smean64"));
        const double wr = use_kv ? (double)ATTN_KV_IMG_BYTES * 90.0 / 96.0 : 90.0 * 256 * 4 * (gather_table ? 2 : 1);
        LAUNCH(c, s, "mocha_instnorm", "dec.in_cha", 0.0, b * (90.0 * 256 * 4 + wr),
               launch_instnorm(gather_table ? gather_table : cha, use_kv ? nullptr : WS(c, "kin"), WS(c, "smean"), nullptr, nullptr, nullptr, b, 90, s, &ex));
        if (gather_table) cha = use_kv ? nullptr : WS(c, "sel");
        int rc = run_style(c, b, s);
        if (rc) return rc;
    }
    const float* gbp = cached ? gb_table : WS(c, "gb");
    const float* x = src;
    float* qb = WS(c, "qkv");
    float* kb = qb + (size_t)M * inner;
    float* vb = kb + (size_t)M * inner;
    for (int l = 0; l < c->cfg.dec_depth; ++l) {
        const std::string p = "dec" + std::to_string(l);
        LAUNCH(c, s, "mocha_adain", "dec.adain", 0.0, b * 90.0 * 256 * 4 * 3,
               launch_adain(x, gbp + (size_t)l * 512, 512 * L, WS(c, "xad"), WS(c, "qin"), b, 90, s, c->adain_closed ? 1 : 0,
                            cached ? gather_idx : nullptr, gather_rows, c->inorm_split_max));
        if (c->fold_decoder && DH == 256) {
            // S_h = IN(x) (Wq_h^T Wk_h) IN(cha)^T and out = sum_h (P_h cha) (Wv_h^T Wo_h^T): with dim_head == dim the key and
            // value projections fold into the query and output weights (exact algebra, net/transformer.py:62-76), so the
            // attention reads IN(cha) / cha directly for every head and two of the four projection GEMMs disappear.
            GemmParams gq = plain(WS(c, "qin"), 256, DW(c, p + ".Wqk"), qb, inner, M, inner, 256);
            gq.a_amax = c->gemm_h2 && !c->amax_idle ? c->amax_in : nullptr;
            GEMM(c, s, "dec.q", gq);
            if (use_kv) {
                AttnKvParams a{qb, WS(c, "ao"), reinterpret_cast<const unsigned short*>(WS(c, "kvimg")), inner, inner, b, H, DH, 90, 90, (float)std::pow((double)DH, -0.5), c->attn_kv_pairs ? 1 : 0};
                LAUNCH(c, s, "mocha_attention_x3_kv<256>", "dec.attn", 4.0 * b * H * 90.0 * 90 * DH, 4.0 * M * 2 * inner + (double)b * ATTN_KV_IMG_BYTES,
                       launch_attention_x3_kv(a, s));
            } else {
            AttnParams a{qb, cached ? kin_table : WS(c, "kin"), cached ? gather_table : cha, WS(c, "ao"), inner, 256, 256, inner, b, H, DH, 90, 90, (float)std::pow((double)DH, -0.5), 0, 0};
            if (cached) { a.kv_idx = gather_idx; a.kv_rows = gather_rows; }      // keys / values of window b = the matched entry's rows, in place
            LAUNCH(c, s, attn_kernel_name(c, DH, (long long)b * H), "dec.attn", 4.0 * b * H * 90.0 * 90 * DH, 4.0 * M * (2 * inner + 2 * 256),
                   attention(c, a, s));
            }
            float* out = (l == c->cfg.dec_depth - 1) ? outp : WS(c, "xa");
            int rc = run_out_ff(c, p, WS(c, "ao"), inner, WS(c, "xad"), M, c->cfg.dec_mlp, out, s, ".Wvo", use_kv ? nullptr : v_amax, &x_amax);
            if (rc) return rc;
            if (l == c->cfg.dec_depth - 1 && amax_use(c, x_amax)) { c->amax_dec_of[c->cur] = outp; c->amax_dec[c->cur] = x_amax; }
            x = out;
            continue;
        }
        GemmParams gq = plain(WS(c, "qin"), 256, DW(c, p + ".Wq"), qb, inner, M, inner, 256);
        GEMM(c, s, "dec.q", gq);
        GemmParams gk = plain(WS(c, "kin"), 256, DW(c, p + ".Wk"), kb, inner, M, inner, 256);
        GEMM(c, s, "dec.k", gk);
        GemmParams gv = plain(cha, 256, DW(c, p + ".Wv"), vb, inner, M, inner, 256);
        GEMM(c, s, "dec.v", gv);
        AttnParams a{qb, kb, vb, WS(c, "ao"), inner, inner, inner, inner, b, H, DH, 90, 90, (float)std::pow((double)DH, -0.5)};
        LAUNCH(c, s, attn_kernel_name(c, DH, (long long)b * H), "dec.attn", 4.0 * b * H * 90.0 * 90 * DH,
               4.0 * M * 4 * inner, attention(c, a, s));
        float* out = (l == c->cfg.dec_depth - 1) ? outp : WS(c, "xa");
        int rc = run_out_ff(c, p, WS(c, "ao"), inner, WS(c, "xad"), M, c->cfg.dec_mlp, out, s);
        if (rc) return rc;
        x = out;
    }
    return 0;
}

// to_mot (model.py:71-80)
int run_to_mot(mocha_ctx* c, const float* tokens, int b, float* Y, hipStream_t s, bool denorm = false) {
    const int V = c->cfg.V, M = b * 90;
    const int nn = (V + 1) * c->cfg.C_in;
    if (denorm && !c->pose_norm) return fail(c, MOCHA_ERR_STATE, "de-normalised output needs mocha_set_pose_norm first");
    // two-plane fp16 engine: LeakyReLU and the column-normalised adjacency mixes (body_front, joint_expand: non-negative coefficients that sum
    // to at most one per output) do not raise the largest magnitude, so each GEMM's bound is its predecessor's output bound
    { int rc = amax_begin(c, 2, s, b); if (rc) return rc; }
    float* t_amax = nullptr;
    if (c->gemm_h2 && !c->amax_idle) {
        if (tokens == c->amax_dec_of[c->cur] && amax_use(c, c->amax_dec[c->cur])) t_amax = c->amax_dec[c->cur];
        else { int rc = amax_measure(c, s, tokens, b, 90ll * 256, &t_amax); if (rc) return rc; }
    }
    LAUNCH(c, s, "mocha_body_front", "mot.body_front", b * 90.0 * 512 * 12, b * 90.0 * (256 + 512) * 4, launch_body_front(tokens, DW(c, "A_b"), WS(c, "xA"), b * 15, s));
    GemmParams g1 = plain(WS(c, "xA"), 512, DW(c, "mot.Wgb"), WS(c, "t1"), 256, M, 256, 512);
    g1.rowbias = DW(c, "mot.rbb"); g1.rb_mod = 6;
    g1.a_amax = amax_use(c, t_amax); g1.c_amax = amax_slot(c);
    GEMM(c, s, "mot.gcn_body", g1);
    GemmParams g2 = plain(WS(c, "t1"), 256, DW(c, "mot.Wtb"), WS(c, "x5"), 256, M, 256, 768);
    g2.gather = 1; g2.T_out = 15; g2.V = 6; g2.ntaps = 3; g2.pad = 1; g2.stride = 1; g2.R = 1; g2.T_full = 15;
    g2.tshift = 0; g2.Cc = 256; g2.T_src = 15; g2.bias = DW(c, "mot.btb");
    g2.a_amax = amax_use(c, g1.c_amax); g2.c_amax = amax_slot(c);
    GEMM(c, s, "mot.tcn_body", g2);
    // joint block gcn conv at body-part resolution (upsample + unpool only copy rows):  lrelu -> 256 -> 3*64
    GemmParams g3 = plain(WS(c, "x5"), 256, DW(c, "mot.Wg2"), WS(c, "g"), 192, M, 192, 256);
    g3.a_lrelu = 1; g3.bias = DW(c, "mot.bg2");
    g3.a_amax = amax_use(c, g2.c_amax); g3.c_amax = amax_slot(c);
    GEMM(c, s, "mot.gcn_joint", g3);
    LAUNCH(c, s, "mocha_joint_expand", "mot.joint_expand", b * 15.0 * V * 64 * 36, b * 15.0 * (6 * 192 + V * 64) * 4, launch_joint_expand(WS(c, "g"), DW(c, "AU"), WS(c, "y2c"), b * 15, V, s));
    // temporal conv k=5 over the x4-upsampled frames, read through the gather (t >> 2)
    if (c->fold_upsample) {
        // rows (window, source frame s, joint), columns (phase, channel): z[(b, s, v)][phase*64 + c] is the conv's output at frame 4 s + phase
        GemmParams g4 = plain(WS(c, "y2c"), 64, DW(c, "mot.Wt2p"), WS(c, "z"), 256, b * 15 * V, 256, 192);
        g4.gather = 1; g4.T_out = 15; g4.V = V; g4.ntaps = 3; g4.pad = 2; g4.stride = 4; g4.tstep = 4; g4.R = 1; g4.T_full = 60;
        g4.tshift = 2; g4.Cc = 64; g4.T_src = 15; g4.bias = DW(c, "mot.bt2p");
        g4.a_amax = amax_use(c, g3.c_amax); g4.rows_per_win = 15 * V;
        if (b >= c->upsample_split_min) {
            // large batches: two launches of two taps each instead of one with a third of its weight blocks zero
            GemmParams ga = g4; ga.W = DW(c, "mot.Wt2a"); ga.N = 128; ga.K = 128; ga.ntaps = 2;
            GEMM(c, s, "mot.tcn_joint", ga);
            GemmParams gb = ga; gb.W = DW(c, "mot.Wt2b"); gb.C = WS(c, "z") + 128; gb.pad = -2; gb.bias = DW(c, "mot.bt2p") + 128;
            GEMM(c, s, "mot.tcn_joint", gb);
        } else
        GEMM(c, s, "mot.tcn_joint", g4);
    } else {
    GemmParams g4 = plain(WS(c, "y2c"), 64, DW(c, "mot.Wt2"), WS(c, "z"), 64, b * 60 * V, 64, 320);
    g4.gather = 1; g4.T_out = 60; g4.V = V; g4.ntaps = 5; g4.pad = 2; g4.stride = 1; g4.R = 1; g4.T_full = 60;
    g4.tshift = 2; g4.Cc = 64; g4.T_src = 15; g4.bias = DW(c, "mot.bt2");
    GEMM(c, s, "mot.tcn_joint", g4);
    }
    LAUNCH(c, s, "mocha_final_proj", "mot.final_proj", b * 60.0 * V * 64 * 15 * 2, b * 60.0 * V * (64 + 15) * 4, launch_final_proj(WS(c, "z"), DW(c, "mot.W6"), DW(c, "mot.b6"), Y, b * 60 * V, c->cfg.C_in, V,
                             denorm ? c->pose_norm + 2 * nn : nullptr, denorm ? c->pose_norm + 3 * nn : nullptr, s, c->fold_upsample ? 1 : 0));
    return 0;
}

// Run fn(b0, b, stream) over the batch.  Small batches: chunk by chunk on the caller's stream.  Large ones: the two
// halves of the batch go to the caller's stream and to `aux` (forked / joined with events, capture-safe), each with its
// own workspace set, so two independent kernel chains are in flight and fill each other's bubbles.
template <class F>
int for_chunks(mocha_ctx* c, int B, hipStream_t s, F&& fn) {
    // set 1 must be a FULL-size set: with lanes >= 2 it exists at 8 windows whenever dual_sets was false at allocation
    const bool dual = c->dual_stream && B >= c->dual_min && c->dual_ready && !c->wss[1].empty();
    if (!dual) {
        c->cur = c->lane;
        for (int b0 = 0; b0 < B; b0 += c->chunk) { int rc = fn(b0, std::min(c->chunk, B - b0), s); if (rc) return rc; }
        return 0;
    }
    HIPCHK(c, hipEventRecord(c->ev_fork, s));
    HIPCHK(c, hipStreamWaitEvent(c->aux, c->ev_fork, 0));
    const int h = (B + 1) / 2;
    int rc = 0;
    c->cur = 0;
    for (int b0 = 0; b0 < h && !rc; b0 += c->chunk) rc = fn(b0, std::min(c->chunk, h - b0), s);
    c->cur = 1;
    for (int b0 = h; b0 < B && !rc; b0 += c->chunk) rc = fn(b0, std::min(c->chunk, B - b0), c->aux);
    c->cur = 0;
    // join even when a half failed: the caller's stream must not be left forked (an open capture would be invalidated)
    const hipError_t ej = hipEventRecord(c->ev_join, c->aux);
    const hipError_t ew = ej == hipSuccess ? hipStreamWaitEvent(s, c->ev_join, 0) : ej;
    if (rc) return rc;
    HIPCHK(c, ew);
    return 0;
}

int ready(mocha_ctx* c, int B) {
    if (!c) return MOCHA_ERR_ARG;
    if (!c->finalized) return fail(c, MOCHA_ERR_STATE, "weights not finalised: call mocha_finalize_weights first");
    if (B < 0) return fail(c, MOCHA_ERR_ARG, "negative batch");
    HIPCHK(c, hipSetDevice(c->device));
    if (B > 0) return ensure_ws(c, B);
    return 0;
}

// required device pointers of a call with B > 0: a NULL must come back as an error code, not as a GPU fault that takes the host down
#define NEED_PTRS(c, B, what, ...)                                                                     \
    do {                                                                                               \
        if ((B) > 0) { const void* need__[] = {__VA_ARGS__}; for (const void* p__ : need__) if (!p__) return fail((c), MOCHA_ERR_ARG, what ": null argument"); } \
    } while (0)

// Nearest bank entry of every query (BallTree.query(k=1), test_fullframework.py:296,443).
// Precision: the rows of one character's bank sit close together far from the origin (||b||^2 ~ 1e5, gaps between the
// best candidates ~ 1e-2), so ||b||^2 - 2 q.b in fp32 cannot rank them.  Few queries: the HBM-bound scan evaluates
// sum (q-b)^2 directly.  Many queries: the GEMM runs on operands centred on the bank centroid c (distances are
// translation invariant; the centred norms are of the size of the distances themselves) and the arg-min re-ranks its
// four best candidates by their exact direct distances.
int grow(mocha_ctx* c, DevBuf& b, size_t need) {
    if (b.n >= need) return 0;
    HIPCHK(c, hipDeviceSynchronize());
    dev_free(c, b.p);
    b = DevBuf{};
    int rc = dev_alloc(c, &b.p, need);
    if (rc) return rc;
    b.n = need;
    c->generation++;
    return 0;
}

static constexpr int64_t SCAN16_MIN = 4096;        // rows from which an fp32 bank is scanned through its bf16 copy (few queries)
static constexpr int64_t X3_BANK_MAX = 4096;       // rows: the packed image of a 4096-row bank (BASELINE configs[2]) is 566 MB

// split of the many-query GEMM's K loop over gridDim.z so that a launch has about three tiles per CU
int match_ksplit(int Q, int64_t N) {
    const long long tiles = (long long)((Q + 127) / 128) * ((N + 127) / 128);
    return (int)std::min<long long>(16, std::max<long long>(1, (768 + tiles - 1) / tiles));
}

// windows workspace set `set` holds (ensure_ws): the chunk for set 0 and dual_stream's set 1, a handful for a lane's set
int set_windows(const mocha_ctx* c, int set) {
    return (set == 0 || (set == 1 && c->dual_ready)) ? std::max(c->chunk, 8) : 8;
}

// Scratch of do_match for up to Q queries against N bank rows, in workspace set `set`: centred queries, the streaming
// scan's partial minima, the GEMM's score slabs.  mocha_reserve and mocha_bank_set call it for the workspace chunk, so
// that a steady-state match never allocates (graph-capture safe); do_match itself calls it for exactly its Q.
int ensure_match_scratch(mocha_ctx* c, int set, int Q, int64_t N, bool every_q_up_to) {
    const size_t D = 90 * 256;
    int rc;
    if ((rc = grow(c, c->match_qc[set], (size_t)std::max(Q, 8) * D))) return rc;
    const size_t need_ws = match_stream_scratch(8, N);
    if (c->best_ws_n[set] < need_ws) {
        HIPCHK(c, hipDeviceSynchronize());
        if (c->best_ws[set]) (void)hipFree(c->best_ws[set]);
        c->best_ws[set] = nullptr; c->best_ws_n[set] = 0;
        void* bp = nullptr;
        HIPCHK(c, hipMalloc(&bp, sizeof(unsigned long long) * need_ws));
        c->best_ws[set] = (unsigned long long*)bp; c->best_ws_n[set] = need_ws;
        c->generation++;
    }
    if (Q > 8) {
        size_t need = 0;
        for (int q = every_q_up_to ? 9 : Q; q <= Q; ++q)
            need = std::max(need, (size_t)std::max(match_ksplit(q, N), match_pass256_ksplit(q, N)) * q * (size_t)N);
        if ((rc = grow(c, c->match_S[set], need))) return rc;
        if ((rc = grow(c, c->match_qstat[set], (size_t)2 * QSTAT_PARTS * std::max(Q, 256)))) return rc;
    }
    if (c->scan16 && !c->bank_is_bf16 && N >= SCAN16_MIN && c->scan_keys_n[set] < (size_t)8 * N) {      // every row's coarse key, 8 queries
        HIPCHK(c, hipDeviceSynchronize());
        if (c->scan_keys[set]) (void)hipFree(c->scan_keys[set]);
        c->scan_keys[set] = nullptr; c->scan_keys_n[set] = 0;
        void* kp = nullptr;
        HIPCHK(c, hipMalloc(&kp, sizeof(unsigned long long) * match_scan16_scratch_words(N)));
        // the head (slice winners, arrival tickets of mocha_match_refine) starts at zero; the tickets only ever count up
        HIPCHK(c, hipMemset(kp, 0, sizeof(unsigned long long) * match_scan16_scratch_head_words()));
        c->scan_keys[set] = (unsigned long long*)kp; c->scan_keys_n[set] = (size_t)8 * N;
        c->generation++;
    }
    return 0;
}

// qc_pre: the queries minus the current bank's centroid, fp32, when the caller's producer already wrote them (mocha_instnorm's zc)
// (fp32; bf16 where the many-query pass against a bf16 bank wants them so: qc_pre_bf16, mocha_instnorm's zc16)
// the round-4 selection (match_select2.hip) serves this many queries against the current bank?
bool use_select2(const mocha_ctx* c, int Q) { return c->select2 && Q > 8 && (!c->bank_is_bf16 || c->match_planes == 2); }

// qstat_pre: with select2 and Q > 8, the producer also left the row statistics in match_qstat[set] (and, bf16: BOTH query planes in qc_pre)
int do_match(mocha_ctx* c, const float* qnm, int Q, int32_t* idx, float* dist, hipStream_t s, const void* qc_pre_any = nullptr,
             bool qc_pre_bf16 = false, bool qstat_pre = false) {
    if (!c->bank_cnt || c->bank_N <= 0) return fail(c, MOCHA_ERR_STATE, "no bank: call mocha_bank_set first");
    const int D = 90 * 256;
    const int64_t N = c->bank_N;
    const int set = c->cur;
    int rc;
    if ((rc = ensure_match_scratch(c, set, Q, N, false))) return rc;
    // centred queries: the bf16 bank holds bf16(b - c) and the GEMMs work on centred operands.  The many-query bf16 pass takes
    // them as bf16 (match_qc holds Q x D bf16 then), everything else as fp32.
    const bool via16 = Q <= 8 && !c->bank_is_bf16 && c->bank16f_valid && c->scan16 && c->scan_keys_n[set] >= (size_t)8 * N;
    const bool need_qc = c->bank_is_bf16 || Q > 8 || via16;
    // the many-query pass against a bf16 bank takes the centred queries as bf16, everything else as fp32: a precomputed copy of the
    // other kind is of no use (mocha_center_bf16 / mocha_sub_rows then make the right one)
    const bool sel2 = use_select2(c, Q);                       // many-query path with the round-4 selection: row statistics from the producer
    const float* qc_pre = (qc_pre_any && qc_pre_bf16 == (c->bank_is_bf16 && Q > 8) && (!sel2 || qstat_pre)) ? static_cast<const float*>(qc_pre_any) : nullptr;
    if (need_qc && !qc_pre) {
        if (sel2 && c->bank_is_bf16)
            LAUNCH(c, s, "mocha_center_rows", "match.center", 0.0, 8.0 * Q * D, launch_center_rows(qnm, c->bank_center, c->match_qc[set].p, c->match_planes, nullptr, c->match_qstat[set].p, Q, D, s));
        else if (sel2)
            LAUNCH(c, s, "mocha_center_rows", "match.center", 0.0, 8.0 * Q * D, launch_center_rows(qnm, c->bank_center, nullptr, 0, c->match_qc[set].p, c->match_qstat[set].p, Q, D, s));
        else if (c->bank_is_bf16 && Q > 8)
            LAUNCH(c, s, "mocha_center_bf16", "match.center", 0.0, 6.0 * Q * D, launch_center_bf16(qnm, c->bank_center, c->match_qc[set].p, Q, D, s));
        else
            LAUNCH(c, s, "mocha_sub_rows", "match.center", 0.0, 8.0 * Q * D, launch_sub_rows(qnm, c->bank_center, c->match_qc[set].p, Q, D, s));
    }
    const float* qc = need_qc ? (qc_pre ? qc_pre : c->match_qc[set].p) : nullptr;
    if (via16) {
        // fp32 bank through its centred bf16 copy: every row's coarse distance from half the bytes, then the exact fp32 distances of
        // the rows the rounding bound cannot exclude - the result of the fp32 search
        if (c->scan8 && c->bank8_valid && Q <= 4) {
            // the adaptive byte stage: 1 B per value while the queries sit close to a few rows, the bf16 scan otherwise (decided on the device)
            LAUNCH(c, s, "mocha_match_scan_adaptive+refine", "match.stream8", 3.0 * Q * N * D, (double)N * D * 1.0 + 4.0 * Q * D + 16.0 * Q * N,
                   launch_match_scan8(c->bank8, c->bank8_scale, c->bank8_rho, c->bank16f, c->bank_rho, c->bank_cnt, qc, qnm, Q, N, D, c->scan_keys[set], idx, dist, s));
            return 0;
        }
        LAUNCH(c, s, "mocha_match_stream<bf16>+refine", "match.stream16", 3.0 * Q * N * D,
               ((Q + 7) / 8) * (double)N * D * 2.0 + 4.0 * Q * D + 16.0 * Q * N,
               launch_match_scan16(c->bank16f, c->bank_rho, c->bank_cnt, qc, qnm, Q, N, D, c->scan_keys[set], idx, dist, s));
        return 0;
    }
    if (Q <= 8) {
        const void* bank = c->bank_is_bf16 ? (const void*)c->bank_bf16 : (const void*)c->bank_cnt;
        const double passes = (Q + 7) / 8;
        LAUNCH(c, s, c->bank_is_bf16 ? "mocha_match_stream<bf16>" : "mocha_match_stream<f32>", "match.stream", 3.0 * Q * N * D,
               passes * N * D * (c->bank_is_bf16 ? 2.0 : 4.0) + 4.0 * Q * D,
               launch_match_stream(bank, c->bank_is_bf16 ? 1 : 0, c->bank_is_bf16 ? qc : qnm, Q, N, D, c->best_ws[set], idx, dist, s));
        return 0;
    }
    DevBuf& mS = c->match_S[set];
    if (c->bank_is_bf16) {
        // bf16 bank: one bf16 plane of the centred queries against the centred bf16 bank; the select kernel re-evaluates every
        // row whose coarse score is within the query rounding's error bound of the best one - 2 ||dq|| (||b|| + ||b0||) with the query's
        // measured rounding residual dq, plus 5e-5 (2||q||^2 + ||b||^2 + ||b0||^2) of slack for the pass's fp32 accumulation - exactly
        // (fp32 centred query against the bf16 rows), so the result is the exact search over the rounded bank
        const int npl = sel2 ? c->match_planes : 1;
        const bool pass256 = c->match_pass && !c->use_tiled && Q <= c->match_pass_max_q;
        const int ksplit = pass256 ? match_pass256_ksplit(Q, N) : match_bf16_ksplit(Q, N);
        const void* tiled = nullptr;
        if (c->use_tiled && (size_t)N * D * 2 <= ((size_t)4 << 30)) {          // up to 4 GB of image (N = 93 k rows)
            const size_t need = match_tiled_elems(N, D);
            if (!c->bank_tiled_valid) {
                if (c->bank_tiled_cap < need) {
                    HIPCHK(c, hipDeviceSynchronize());
                    if (c->bank_tiled) (void)hipFree(c->bank_tiled);
                    c->bank_tiled = nullptr; c->bank_tiled_cap = 0;
                    HIPCHK(c, hipMalloc(&c->bank_tiled, need * 2));
                    c->bank_tiled_cap = need;
                }
                LAUNCH(c, s, "mocha_tile_bf16", "bank.tile", 0.0, 4.0 * N * D, launch_tile_bf16(c->bank_bf16, c->bank_tiled, N, D, s));
                c->bank_tiled_valid = true;
            }
            tiled = c->bank_tiled;
        }
        if (pass256) {
            const void* t32 = nullptr;
            if (c->match_pass == 2 && (size_t)N * D * 2 <= ((size_t)4 << 30)) {
                const size_t need = match_tile32_elems(N, D);
                if (!c->bank_tile32_valid) {
                    if (c->bank_tile32_cap < need) {
                        HIPCHK(c, hipDeviceSynchronize());
                        if (c->bank_tile32) (void)hipFree(c->bank_tile32);
                        c->bank_tile32 = nullptr; c->bank_tile32_cap = 0;
                        HIPCHK(c, hipMalloc(&c->bank_tile32, need * 2));
                        c->bank_tile32_cap = need;
                    }
                    LAUNCH(c, s, "mocha_tile32_bf16", "bank.tile32", 0.0, 4.0 * N * D, launch_tile32_bf16(c->bank_bf16, c->bank_tile32, N, D, s));
                    c->bank_tile32_valid = true;
                }
                t32 = c->bank_tile32;
            }
            LAUNCH(c, s, "mocha_match_pass256", "match.qk_bf16", 2.0 * npl * Q * (double)N * D, 2.0 * npl * Q * D + 2.0 * N * D + 4.0 * Q * N * ksplit,
                   launch_match_pass256(qc, c->bank_bf16, mS.p, Q, N, D, ksplit, s, c->match_pass_variant, npl, t32));
        } else
        {
        unsigned* tk = nullptr;
        if (c->match_fold && ksplit > 1) {
            const size_t need = (size_t)((Q + 127) / 128) * (size_t)((N + 127) / 128);
            if (c->fold_tickets_n[set] < need) {
                HIPCHK(c, hipDeviceSynchronize());
                if (c->fold_tickets[set]) (void)hipFree(c->fold_tickets[set]);
                c->fold_tickets[set] = nullptr; c->fold_tickets_n[set] = 0;
                void* tp = nullptr;
                HIPCHK(c, hipMalloc(&tp, need * sizeof(unsigned)));
                HIPCHK(c, hipMemset(tp, 0, need * sizeof(unsigned)));
                c->fold_tickets[set] = (unsigned*)tp; c->fold_tickets_n[set] = need;
                c->generation++;
            }
            tk = c->fold_tickets[set];
        }
        LAUNCH(c, s, "mocha_match_gemm_bf16", "match.qk_bf16", 2.0 * npl * Q * (double)N * D, 2.0 * npl * Q * D + 2.0 * N * D + 4.0 * Q * N * ksplit,
               launch_match_gemm_bf16(qc, c->bank_bf16, mS.p, Q, N, D, ksplit, s, npl, tiled, (c->match_nt && Q <= 128) ? 1 : 0, tk));
        if (tk) {                                        // slab 0 holds the sum: the selection reads one slab
            if (sel2)
                LAUNCH(c, s, "mocha_match_select2", "match.select", 0.0, 4.0 * Q * N + 4.0 * N + 8.0 * Q,
                       launch_match_select2(mS.p, 1, (long long)Q * N, (int)N, c->bank_norm, qnm, c->bank_center, nullptr, c->bank_bf16,
                                            c->match_qstat[set].p, 5e-5f, Q, N, D, idx, dist, s));
            else
                LAUNCH(c, s, "mocha_match_select", "match.select", 0.0, 4.0 * Q * N + Q * (8.0 * D + 8 * 2.0 * D),
                       launch_match_select(mS.p, 1, (long long)Q * N, (int)N, c->bank_norm, qnm, c->bank_center, nullptr, c->bank_bf16, 5e-5f,
                                           Q, N, D, idx, dist, s));
            return 0;
        }
        }
        if (sel2)
            LAUNCH(c, s, "mocha_match_select2", "match.select", 0.0, 4.0 * ksplit * Q * N + 4.0 * N + 8.0 * Q,
                   launch_match_select2(mS.p, ksplit, (long long)Q * N, (int)N, c->bank_norm, qnm, c->bank_center, nullptr, c->bank_bf16,
                                        c->match_qstat[set].p, 5e-5f, Q, N, D, idx, dist, s));
        else
        LAUNCH(c, s, "mocha_match_select", "match.select", 0.0, 4.0 * ksplit * Q * N + Q * (8.0 * D + 8 * 2.0 * D),
               launch_match_select(mS.p, ksplit, (long long)Q * N, (int)N, c->bank_norm, qnm, c->bank_center, nullptr, c->bank_bf16, 5e-5f,
                                   Q, N, D, idx, dist, s));
        return 0;
    }
    const int ksplit = match_ksplit(Q, N);
    GemmParams g = plain(qc, D, c->bank_cnt, mS.p, (int)N, Q, (int)N, D);
    g.ksplit = ksplit; g.slab_stride = (long long)Q * N;
    if (c->bank_x3_valid && c->gemm_x3) {
        // the plane engine wants an even number of K steps in every slab: the largest split <= ksplit that divides 23040 / 32
        int kx = ksplit;
        while (kx > 1 && (D / 32) % kx != 0) --kx;
        g.ksplit = kx;
    }
    if (c->bank_x3_valid && c->gemm_x3 && gemm_x3_supports(g)) {
        g.Wsplit = c->bank_x3;                          // the centred bank as planes (bank_set_impl); same coarse-score accuracy as below
        g.persistent = c->gemm_persistent; g.persistent_max_n = c->gemm_persistent_max_n;
        LAUNCH(c, s, "mocha_gemm_x3", "match.qk", 2.0 * Q * (double)N * D, 4.0 * ((double)Q * D + (double)Q * N * g.ksplit) + 6.0 * N * D,
               launch_gemm_x3(g, s));
    } else {
        g.ksplit = ksplit;
        g.wsub = c->bank_center;
        GEMM(c, s, "match.qk", g);
    }
    const int nslab = g.ksplit;
    // exact-f32 MFMA on centred operands: a coarse score is accurate to ~4e-7 (||q-c||^2 + ||b-c||^2); candidates within ten
    // times that of the best are re-evaluated in the direct form
    if (sel2)
        LAUNCH(c, s, "mocha_match_select2", "match.select", 0.0, 4.0 * nslab * Q * N + 4.0 * N + 8.0 * Q,
               launch_match_select2(mS.p, nslab, (long long)Q * N, (int)N, c->bank_norm, qnm, c->bank_center, c->bank_cnt, nullptr,
                                    c->match_qstat[set].p, 4e-6f, Q, N, D, idx, dist, s));
    else
    LAUNCH(c, s, "mocha_match_select", "match.select", 0.0, 4.0 * nslab * Q * N + Q * 16.0 * D,
           launch_match_select(mS.p, nslab, (long long)Q * N, (int)N, c->bank_norm, qnm, c->bank_center, c->bank_cnt, nullptr, 4e-6f,
                               Q, N, D, idx, dist, s));
    return 0;
}

}  // namespace

// =========================================================================================== C ABI
extern "C" {

int mocha_abi_version(void) { return 5; }

#define MOCHA_STR2(x) #x
#define MOCHA_STR(x) MOCHA_STR2(x)
const char* mocha_build_info(void) {
    return "hipcc HIP " MOCHA_STR(HIP_VERSION_MAJOR) "." MOCHA_STR(HIP_VERSION_MINOR) "." MOCHA_STR(HIP_VERSION_PATCH) "-" HIP_VERSION_GITHASH " gfx950";
}
int mocha_runtime_version(void) {
    int v = 0;
    return hipRuntimeGetVersion(&v) == hipSuccess ? v : -1;
}

int64_t mocha_generation(const mocha_ctx* c) { return c ? c->generation : 0; }

const char* mocha_last_error(const mocha_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

int mocha_create(const mocha_cfg* cfg, int device, mocha_ctx** out) {
    if (!cfg || !out) return fail(nullptr, MOCHA_ERR_ARG, "null argument");
    *out = nullptr;
    // the kernels are specialised for the shipped architecture (configs/config.yaml:13-31)
    if (cfg->T != 60 || cfg->patch != 4 || cfg->dim != 256 || cfg->C_in != 15)
        return fail(nullptr, MOCHA_ERR_ARG, "unsupported T/patch/dim/C_in (%d/%d/%d/%d), need 60/4/256/15", cfg->T, cfg->patch, cfg->dim, cfg->C_in);
    // head dims: 64 (the reference Attention's own default, net/transformer.py:38: exact-f32 MFMA attention), 128 and 256 (also the plane engine)
    for (int dh : {cfg->enc_dim_head, cfg->dec_dim_head})
        if (dh != 64 && dh != 128 && dh != 256) return fail(nullptr, MOCHA_ERR_ARG, "dim_head must be 64, 128 or 256 (got %d)", dh);
    if (cfg->enc_heads * cfg->enc_dim_head > 1024 || cfg->dec_heads * cfg->dec_dim_head > 1024 || cfg->enc_heads < 1 || cfg->dec_heads < 1)
        return fail(nullptr, MOCHA_ERR_ARG, "heads*dim_head must be in [64, 1024]");
    for (int mlp : {cfg->enc_mlp, cfg->dec_mlp})
        if (mlp < 64 || mlp > 2048 || mlp % 64) return fail(nullptr, MOCHA_ERR_ARG, "mlp_dim must be a multiple of 64 in [64, 2048] (got %d)", mlp);
    if (cfg->enc_depth < 1 || cfg->enc_depth > 8 || cfg->dec_depth < 1 || cfg->dec_depth > 8) return fail(nullptr, MOCHA_ERR_ARG, "depth must be in [1, 8]");
    mocha_ctx* c = new mocha_ctx();
    c->cfg = *cfg; c->device = device;
    if (!make_skeleton(cfg->layout, c->sk) || c->sk.V != cfg->V) {
        int v = c->sk.V; delete c;
        return fail(nullptr, MOCHA_ERR_ARG, "layout %d has %d joints, cfg.V = %d (0 = mocha/24, 1 = mixamo/22)", cfg->layout, v, cfg->V);
    }
    build_expectations(c);
    hipError_t e = hipSetDevice(device);
    if (e == hipSuccess) e = gemm_init();
    if (e == hipSuccess) e = gemm_x3_init();
    if (e == hipSuccess) e = gemm_h2_init();
    if (e == hipSuccess) e = gemm_x3r_init();
    if (e == hipSuccess) e = match_scan8_init();
    if (e == hipSuccess) e = attention_x3_init();
    if (e == hipSuccess) e = match_mfma_init();
    if (e == hipSuccess) e = match_refine_init();
    if (e == hipSuccess) e = featurize_init();
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->aux, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming);
    if (e != hipSuccess) { delete c; return fail(nullptr, MOCHA_ERR_HIP, "device %d init failed: %s", device, hipGetErrorString(e)); }
    *out = c;
    return 0;
}

void mocha_destroy(mocha_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    for (float* p : c->owned) (void)hipFree(p);
    for (int set = 0; set < mocha_ctx::MAX_SETS; ++set) { if (c->idx_ws[set]) (void)hipFree(c->idx_ws[set]); if (c->best_ws[set]) (void)hipFree(c->best_ws[set]); }
    if (c->bank_bf16) (void)hipFree(c->bank_bf16);
    if (c->bank_tiled) (void)hipFree(c->bank_tiled);
    if (c->bank_tile32) (void)hipFree(c->bank_tile32);
    if (c->bank_x3) (void)hipFree(c->bank_x3);
    if (c->pair_x3) (void)hipFree(c->pair_x3);
    if (c->bank16f) (void)hipFree(c->bank16f);
    if (c->bank8) (void)hipFree(c->bank8);
    for (auto* k : c->fold_tickets) if (k) (void)hipFree(k);
    for (auto* k : c->scan_keys) if (k) (void)hipFree(k);
    if (c->bcast_hdr) (void)hipFree(c->bcast_hdr);
    if (c->topk_keys) (void)hipFree(c->topk_keys);
    if (c->aux) (void)hipStreamDestroy(c->aux);
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    if (c->ev_join) (void)hipEventDestroy(c->ev_join);
    if (c->bone_parents) (void)hipFree(c->bone_parents);
    for (auto& g : c->step) {
        if (g.exec) (void)hipGraphExecDestroy(g.exec);
        if (g.graph) (void)hipGraphDestroy(g.graph);
    }
    if (c->cap_stream) (void)hipStreamDestroy(c->cap_stream);
    (void)mocha_comm_destroy(c);
    delete c;
}

int mocha_load_weight(mocha_ctx* c, const char* name, const float* host, const int64_t* shape, int ndim) {
    if (!c || !name || !host || !shape) return fail(c, MOCHA_ERR_ARG, "null argument");
    const std::string n(name);
    if (is_graph_buffer(n)) return check_graph_buffer(c, n, host, shape, ndim);
    auto it = c->expect.find(n);
    if (it == c->expect.end()) return fail(c, MOCHA_ERR_WEIGHT, "unknown weight name '%s'", name);
    if ((int)it->second.size() != ndim) return fail(c, MOCHA_ERR_WEIGHT, "%s: rank %d, expected %zu", name, ndim, it->second.size());
    size_t cnt = 1;
    for (int i = 0; i < ndim; ++i) {
        if (shape[i] != it->second[i]) return fail(c, MOCHA_ERR_WEIGHT, "%s: dim %d is %lld, expected %lld", name, i, (long long)shape[i], (long long)it->second[i]);
        cnt *= (size_t)shape[i];
    }
    HostTensor t; t.data.assign(host, host + cnt); t.shape.assign(shape, shape + ndim);
    c->host_w[n] = std::move(t);
    c->finalized = false;
    return 0;
}

int mocha_finalize_weights(mocha_ctx* c) {
    if (!c) return MOCHA_ERR_ARG;
    for (auto& kv : c->expect)
        if (!c->host_w.count(kv.first)) return fail(c, MOCHA_ERR_STATE, "missing weight '%s'", kv.first.c_str());
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipDeviceSynchronize());
    x3_drop_images(c);                              // re-loaded weights keep their device addresses
    h2_drop_images(c);
    const Skeleton& sk = c->sk;
    const int V = sk.V;
    int rc = 0;
    auto up = [&](const std::string& n, const std::vector<float>& v) { if (!rc) rc = upload(c, n, v); };
    auto f32 = [](const std::vector<double>& v) { return std::vector<float>(v.begin(), v.end()); };

    // ---- graph operators
    // AP[k][v][p] = sum_w A_j[k][v][w] pool[w][p]   (adjacency then joint->part mean, graph.py:463-465)
    std::vector<double> AP((size_t)3 * V * 6, 0.0), APsum(3 * 6, 0.0);
    for (int k = 0; k < 3; ++k)
        for (int v = 0; v < V; ++v)
            for (int p = 0; p < 6; ++p) {
                double a = 0;
                for (int w = 0; w < V; ++w) a += (double)(float)sk.A_j[((size_t)k * V + v) * V + w] * sk.pool[(size_t)w * 6 + p];
                AP[((size_t)k * V + v) * 6 + p] = a;
                APsum[k * 6 + p] += a;
            }
    up("AP", f32(AP));
    // AU[k][p][w] = sum_{v in part p} A_j[k][v][w]    (part->joint copy then adjacency, graph.py:606-608)
    std::vector<double> AU((size_t)3 * 6 * V, 0.0);
    for (int k = 0; k < 3; ++k)
        for (int v = 0; v < V; ++v)
            for (int w = 0; w < V; ++w) AU[((size_t)k * 6 + sk.part_of[v]) * V + w] += (double)(float)sk.A_j[((size_t)k * V + v) * V + w];
    up("AU", f32(AU));
    up("A_b", f32(sk.A_b));
    double Absum[2 * 6];
    for (int k = 0; k < 2; ++k)
        for (int w = 0; w < 6; ++w) { double a = 0; for (int v = 0; v < 6; ++v) a += (double)(float)sk.A_b[(k * 6 + v) * 6 + w]; Absum[k * 6 + w] = a; }

    // ---- mot_embedding
    up("pos_emb", W(c, "pos_emb"));
    up("emb.W1", W(c, "mot_embedding.1.weight"));
    up("emb.b1", W(c, "mot_embedding.1.bias"));
    {
        const auto& w1 = W(c, "mot_embedding.1.weight"); const auto& b1 = W(c, "mot_embedding.1.bias");
        const size_t cin = w1.size() / b1.size();
        double l1 = 0, bm = 0;
        for (size_t o = 0; o < b1.size(); ++o) {
            double a = 0;
            for (size_t i = 0; i < cin; ++i) a += std::fabs((double)w1[o * cin + i]);
            l1 = std::max(l1, a); bm = std::max(bm, std::fabs((double)b1[o]));
        }
        c->emb_l1 = (float)(l1 * (1.0 + 1e-6)); c->emb_bmax = (float)(bm * (1.0 + 1e-6));       // rounded up: a bound
    }
    up("emb.Wg", repack_gcn_adjfirst(W(c, "mot_embedding.2.blk.gcn.conv.weight"), 3, 256, 64));
    {
        const auto& bg = W(c, "mot_embedding.2.blk.gcn.conv.bias");
        std::vector<float> rb(6 * 256);
        for (int p = 0; p < 6; ++p)
            for (int co = 0; co < 256; ++co) { double a = 0; for (int k = 0; k < 3; ++k) a += (double)bg[k * 256 + co] * APsum[k * 6 + p]; rb[p * 256 + co] = (float)a; }
        up("emb.rbg", rb);
    }
    up("emb.Wt", repack_tcn(W(c, "mot_embedding.2.blk.tcn.weight"), 256, 256, 5));
    up("emb.bt", W(c, "mot_embedding.2.blk.tcn.bias"));
    {
        // The joint block applies its 1x1 gcn conv and its k=5 temporal conv back to back (net/blocks.py:126-134: norm and
        // activation act on the block's INPUT, nothing sits between gcn and tcn), so the two compose into one linear map on the
        // pooled, adjacency-mixed 3 x 64 channels: per tap  Wc_tap = Wt_tap (256 x 256) · Wg (256 x 192),  K = 5 x 192 = 960
        // instead of a K = 192 GEMM at 4x the rows plus a K = 1280 one.  fp64, rounded once.
        //   Wc[co][tap*192 + j] = sum_c Wt[co][tap*256 + c] Wg[c][j];   rbc[p][co] = bt[co] + sum_tap sum_c Wt[co][tap*256 + c] rbg[p][c]
        // (the gcn bias is constant over frames, so reflect padding and the 4-frame mean leave it a per-part constant)
        const std::vector<float> Wt = repack_tcn(W(c, "mot_embedding.2.blk.tcn.weight"), 256, 256, 5);
        const std::vector<float> Wg = repack_gcn_adjfirst(W(c, "mot_embedding.2.blk.gcn.conv.weight"), 3, 256, 64);
        const auto& bg = W(c, "mot_embedding.2.blk.gcn.conv.bias");
        const auto& bt = W(c, "mot_embedding.2.blk.tcn.bias");
        std::vector<double> Wc((size_t)256 * 960, 0.0), rbc(6 * 256, 0.0), rbg(6 * 256, 0.0);
        for (int p = 0; p < 6; ++p)
            for (int cc = 0; cc < 256; ++cc) { double a = 0; for (int k = 0; k < 3; ++k) a += (double)bg[k * 256 + cc] * APsum[k * 6 + p]; rbg[p * 256 + cc] = a; }
        for (int co = 0; co < 256; ++co)
            for (int tap = 0; tap < 5; ++tap) {
                double* row = &Wc[(size_t)co * 960 + tap * 192];
                for (int cc = 0; cc < 256; ++cc) {
                    const double wt = Wt[(size_t)co * 1280 + tap * 256 + cc];
                    const float* g = &Wg[(size_t)cc * 192];
                    for (int j = 0; j < 192; ++j) row[j] += wt * (double)g[j];
                    for (int p = 0; p < 6; ++p) rbc[p * 256 + co] += wt * rbg[p * 256 + cc];
                }
            }
        for (int p = 0; p < 6; ++p) for (int co = 0; co < 256; ++co) rbc[p * 256 + co] += (double)bt[co];
        up("emb.Wc", f32(Wc)); up("emb.rbc", f32(rbc));
    }
    for (int which = 0; which < 2; ++which) {
        const std::string src = which ? "to_mot.1.blk." : "mot_embedding.5.blk.";
        const std::string dst = which ? "mot." : "emb.";
        up(dst + "Wgb", repack_gcn_adjfirst(W(c, src + "gcn.conv.weight"), 2, 256, 256));
        const auto& bg = W(c, src + "gcn.conv.bias");
        std::vector<float> rb(6 * 256);
        for (int w = 0; w < 6; ++w)
            for (int co = 0; co < 256; ++co) { double a = 0; for (int k = 0; k < 2; ++k) a += (double)bg[k * 256 + co] * Absum[k * 6 + w]; rb[w * 256 + co] = (float)a; }
        up(dst + "rbb", rb);
        up(dst + "Wtb", repack_tcn(W(c, src + "tcn.weight"), 256, 256, 3));
        up(dst + "btb", W(c, src + "tcn.bias"));
    }
    // ---- transformers
    for (int l = 0; l < c->cfg.enc_depth; ++l) {
        const std::string s = "encoder.layers." + std::to_string(l), d = "enc" + std::to_string(l);
        std::vector<float> qkv = W(c, s + ".1.to_q.1.weight");
        const auto& k = W(c, s + ".1.to_k.1.weight"); const auto& v = W(c, s + ".1.to_v.weight");
        qkv.insert(qkv.end(), k.begin(), k.end()); qkv.insert(qkv.end(), v.begin(), v.end());
        up(d + ".Wqkv", qkv);
        up(d + ".Wo", W(c, s + ".1.to_out.0.weight")); up(d + ".bo", W(c, s + ".1.to_out.0.bias"));
        up(d + ".W1", W(c, s + ".2.net.0.weight")); up(d + ".b1", W(c, s + ".2.net.0.bias"));
        up(d + ".W2", W(c, s + ".2.net.3.weight")); up(d + ".b2", W(c, s + ".2.net.3.bias"));
    }
    for (int l = 0; l < c->cfg.dec_depth; ++l) {
        const std::string s = "decoder.layers." + std::to_string(l), d = "dec" + std::to_string(l);
        up(d + ".Ws2", W(c, s + ".0.style.4.weight")); up(d + ".bs2", W(c, s + ".0.style.4.bias"));      // per layer, for large batches
        up(d + ".Wq", W(c, s + ".1.to_q.1.weight")); up(d + ".Wk", W(c, s + ".1.to_k.1.weight")); up(d + ".Wv", W(c, s + ".1.to_v.weight"));
        up(d + ".Wo", W(c, s + ".1.to_out.0.weight")); up(d + ".bo", W(c, s + ".1.to_out.0.bias"));
        up(d + ".W1", W(c, s + ".2.net.0.weight")); up(d + ".b1", W(c, s + ".2.net.0.bias"));
        up(d + ".W2", W(c, s + ".2.net.3.weight")); up(d + ".b2", W(c, s + ".2.net.3.bias"));
        const int H = c->cfg.dec_heads, DH = c->cfg.dec_dim_head, D = 256;
        if (DH == D) {
            // folded projections (see run_decoder), accumulated in fp64 and rounded once:
            //   Wqk[h*D + j][i] = sum_d Wq[h*DH + d][i] * Wk[h*DH + d][j]        q' = IN(x) Wqk^T, keys = IN(cha)
            //   Wvo[c][h*D + j] = sum_d Wo[c][h*DH + d] * Wv[h*DH + d][j]        out = [P_h cha]_h Wvo^T + bo
            const auto& Wq = W(c, s + ".1.to_q.1.weight"); const auto& Wk = W(c, s + ".1.to_k.1.weight");
            const auto& Wv = W(c, s + ".1.to_v.weight");   const auto& Wo = W(c, s + ".1.to_out.0.weight");
            std::vector<double> qk((size_t)H * D * D, 0.0), vo((size_t)D * H * D, 0.0);
            for (int h = 0; h < H; ++h)
                for (int dd = 0; dd < DH; ++dd) {
                    const float* q = &Wq[(size_t)(h * DH + dd) * D];
                    const float* k = &Wk[(size_t)(h * DH + dd) * D];
                    const float* v = &Wv[(size_t)(h * DH + dd) * D];
                    for (int j = 0; j < D; ++j) {
                        double* row = &qk[((size_t)h * D + j) * D];
                        const double kj = k[j];
                        for (int i = 0; i < D; ++i) row[i] += kj * (double)q[i];
                    }
                    for (int co = 0; co < D; ++co) {
                        double* row = &vo[(size_t)co * H * D + (size_t)h * D];
                        const double w = Wo[(size_t)co * H * DH + h * DH + dd];
                        for (int j = 0; j < D; ++j) row[j] += w * (double)v[j];
                    }
                }
            up(d + ".Wqk", f32(qk)); up(d + ".Wvo", f32(vo));
        }
    }
    {
        // the style MLPs of ALL decoder layers read the same input (the token mean of cha): one GEMM for every layer's first linear
        // (weights stacked), one for the second (block-diagonal: layer l's 512 outputs see only layer l's hidden 512) - two launches
        // instead of two per layer
        const int L = c->cfg.dec_depth;
        std::vector<float> w1((size_t)L * 512 * 256), b1v((size_t)L * 512), w2((size_t)L * 512 * L * 512, 0.f), b2v((size_t)L * 512);
        for (int l = 0; l < L; ++l) {
            const std::string sl = "decoder.layers." + std::to_string(l);
            const auto& a1 = W(c, sl + ".0.style.2.weight"); const auto& c1 = W(c, sl + ".0.style.2.bias");
            const auto& a2 = W(c, sl + ".0.style.4.weight"); const auto& c2 = W(c, sl + ".0.style.4.bias");
            std::copy(a1.begin(), a1.end(), w1.begin() + (size_t)l * 512 * 256);
            std::copy(c1.begin(), c1.end(), b1v.begin() + (size_t)l * 512);
            std::copy(c2.begin(), c2.end(), b2v.begin() + (size_t)l * 512);
            for (int o = 0; o < 512; ++o)
                std::copy(a2.begin() + (size_t)o * 512, a2.begin() + (size_t)(o + 1) * 512, w2.begin() + ((size_t)l * 512 + o) * L * 512 + (size_t)l * 512);
        }
        up("dec.Ws1_all", w1); up("dec.bs1_all", b1v); up("dec.Ws2_blk", w2); up("dec.bs2_all", b2v);
        // ... and the same MLP's operands in float64 (mocha_linear_f64; the fp32 values are exact in float64): first linear stacked over the
        // layers, second linear per layer [L][512][512]
        std::vector<double> w2d((size_t)L * 512 * 512);
        for (int l = 0; l < L; ++l) {
            const auto& a2 = W(c, "decoder.layers." + std::to_string(l) + ".0.style.4.weight");
            std::copy(a2.begin(), a2.end(), w2d.begin() + (size_t)l * 512 * 512);
        }
        auto up64 = [&](const std::string& n, const std::vector<double>& v) {
            if (rc) return;
            double*& d = c->w64[n];
            if (!d) { float* f = nullptr; rc = dev_alloc(c, &f, 2 * v.size()); d = reinterpret_cast<double*>(f); }
            if (!rc && hipMemcpy(d, v.data(), v.size() * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) rc = fail(c, MOCHA_ERR_HIP, "upload of %s failed", n.c_str());
        };
        up64("dec.Ws1_all", std::vector<double>(w1.begin(), w1.end())); up64("dec.bs1_all", std::vector<double>(b1v.begin(), b1v.end()));
        up64("dec.Ws2_all", w2d); up64("dec.bs2_all", std::vector<double>(b2v.begin(), b2v.end()));
        c->bank_dec_valid = false;                          // cached bank constants were made with the previous weights
    }
    // ---- to_mot joint block + head
    up("mot.Wg2", W(c, "to_mot.4.blk.gcn.conv.weight")); up("mot.bg2", W(c, "to_mot.4.blk.gcn.conv.bias"));
    up("mot.Wt2", repack_tcn(W(c, "to_mot.4.blk.tcn.weight"), 64, 64, 5)); up("mot.bt2", W(c, "to_mot.4.blk.tcn.bias"));
    {
        // Output frame t = 4 s + phase of the k = 5 conv over the nearest-x4-upsampled frames (model.py:74, net/blocks.py:112-118) reads
        // upsampled frames t-2 .. t+2 = source frames s-1, s, s+1 only, each several times: the taps that land on one source frame are
        // summed into one weight.  [phase*64 + co][j*64 + ci], j = source frame s-1, s, s+1; a third of the blocks are zero.
        // The reflection at the upsampled ends (frames -2, -1 -> 2, 1; 60, 61 -> 58, 57) stays inside the first / last source frame.
        const std::vector<float>& w = W(c, "to_mot.4.blk.tcn.weight");          // (64, 64, 5, 1)
        static const int JOF[4][5] = {{0, 0, 1, 1, 1}, {0, 1, 1, 1, 1}, {1, 1, 1, 1, 2}, {1, 1, 1, 2, 2}};
        std::vector<double> acc((size_t)256 * 192, 0.0);
        for (int ph = 0; ph < 4; ++ph)
            for (int co = 0; co < 64; ++co)
                for (int ci = 0; ci < 64; ++ci)
                    for (int t = 0; t < 5; ++t) acc[((size_t)ph * 64 + co) * 192 + JOF[ph][t] * 64 + ci] += (double)w[((size_t)co * 64 + ci) * 5 + t];
        std::vector<float> wp(acc.size()), bp(256);
        for (size_t i = 0; i < acc.size(); ++i) wp[i] = (float)acc[i];
        const std::vector<float>& bt = W(c, "to_mot.4.blk.tcn.bias");
        for (int i = 0; i < 256; ++i) bp[i] = bt[i & 63];
        up("mot.Wt2p", wp); up("mot.bt2p", bp);
        // the same without the zero blocks, for large batches: phases 0, 1 read source frames (s-1, s), phases 2, 3 read (s, s+1)
        std::vector<float> wa((size_t)128 * 128), wb((size_t)128 * 128);
        for (int r = 0; r < 128; ++r)
            for (int k = 0; k < 128; ++k) {
                wa[(size_t)r * 128 + k] = wp[(size_t)r * 192 + k];
                wb[(size_t)r * 128 + k] = wp[(size_t)(128 + r) * 192 + 64 + k];
            }
        up("mot.Wt2a", wa); up("mot.Wt2b", wb);
    }
    up("mot.W6", W(c, "to_mot.6.weight")); up("mot.b6", W(c, "to_mot.6.bias"));
    if (rc) return rc;
    c->finalized = true;
    return 0;
}

int mocha_reserve(mocha_ctx* c, int max_batch) {
    if (!c || max_batch < 1) return fail(c, MOCHA_ERR_ARG, "max_batch must be >= 1");
    HIPCHK(c, hipSetDevice(c->device));
    c->max_chunk = max_batch;
    if (c->chunk > max_batch) {           // shrink: drop the larger arena first
        HIPCHK(c, hipDeviceSynchronize());
        c->chunk = 0;
    }
    int rc = ensure_ws(c, max_batch);
    if (rc) return rc;
    if (c->bank_N > 0)                                    // ... and the match scratch of the current bank for that many queries
        for (int set = 0; set < mocha_ctx::MAX_SETS; ++set)
            if (!c->wss[set].empty() && (rc = ensure_match_scratch(c, set, set_windows(c, set), c->bank_N, true))) return rc;
    return 0;
}

int mocha_pos_emb(mocha_ctx* c, const float** dev_ptr) {
    if (!c || !dev_ptr) return MOCHA_ERR_ARG;
    if (!c->finalized) return fail(c, MOCHA_ERR_STATE, "weights not finalised");
    *dev_ptr = DW(c, "pos_emb");
    return 0;
}

int mocha_graph_constants(mocha_ctx* c, float* A_j, float* A_b, float* pool, float* unpool) {
    if (!c) return MOCHA_ERR_ARG;
    const Skeleton& sk = c->sk;
    if (A_j) for (size_t i = 0; i < sk.A_j.size(); ++i) A_j[i] = (float)sk.A_j[i];
    if (A_b) for (size_t i = 0; i < sk.A_b.size(); ++i) A_b[i] = (float)sk.A_b[i];
    if (pool) for (size_t i = 0; i < sk.pool.size(); ++i) pool[i] = (float)sk.pool[i];
    if (unpool) for (size_t i = 0; i < sk.unpool.size(); ++i) unpool[i] = (float)sk.unpool[i];
    return 0;
}

int mocha_embed(mocha_ctx* c, const float* X, int B, float* tokens, int add_pos, void* stream) {
    int rc = ready(c, B); if (rc) return rc;
    NEED_PTRS(c, B, "mocha_embed", X, tokens);
    const size_t xs = (size_t)60 * c->cfg.V * c->cfg.C_in, ts = 90 * 256;
    return for_chunks(c, B, (hipStream_t)stream, [&](int b0, int b, hipStream_t s) -> int {
        return run_embed(c, X + b0 * xs, b, tokens + b0 * ts, add_pos != 0, s);
    });
}

int mocha_encoder(mocha_ctx* c, const float* tokens, int B, float* encoded, void* stream) {
    int rc = ready(c, B); if (rc) return rc;
    NEED_PTRS(c, B, "mocha_encoder", tokens, encoded);
    const size_t ts = 90 * 256;
    for (int i = 0; i < mocha_ctx::MAX_SETS; ++i) c->amax_tok_of[i] = nullptr;      // caller-supplied tokens (the reference adds pos_emb itself): measured, not carried
    return for_chunks(c, B, (hipStream_t)stream, [&](int b0, int b, hipStream_t s) -> int {
        return run_encoder(c, tokens + b0 * ts, b, encoded + b0 * ts, s);
    });
}

int mocha_mvn(mocha_ctx* c, const float* encoded, int B, float* cnt, const float* cnt_mean, const float* cnt_std,
              float* cnt_nm, void* stream) {
    if (!c) return MOCHA_ERR_ARG;                      // needs no weights: usable on a bare context
    if (B == 0) return 0;                              // empty batch: nothing to do, pointers may be null
    if (!encoded || (!cnt && !cnt_nm) || B < 0) return fail(c, MOCHA_ERR_ARG, "bad mvn arguments");
    if (cnt_nm && (!cnt_mean || !cnt_std)) return fail(c, MOCHA_ERR_ARG, "mocha_mvn: cnt_nm needs cnt_mean and cnt_std");
    HIPCHK(c, hipSetDevice(c->device));
    const bool zn = cnt_nm != nullptr;
    hipStream_t s = (hipStream_t)stream;
    const InormExtra iex0 = IEX(c);
    LAUNCH(c, s, "mocha_instnorm", "mvn", 0.0, B * 90.0 * 256 * 4 * (1 + (cnt ? 1 : 0) + (zn ? 1 : 0)), launch_instnorm(encoded, cnt, nullptr, zn ? cnt_mean : nullptr, zn ? cnt_std : nullptr, zn ? cnt_nm : nullptr, B, 90, s, &iex0));
    return 0;
}

int mocha_encode(mocha_ctx* c, const float* X, int B, float* encoded, float* cnt, const float* cnt_mean,
                 const float* cnt_std, float* cnt_nm, void* stream) {
    int rc = ready(c, B); if (rc) return rc;
    NEED_PTRS(c, B, "mocha_encode", X, encoded);
    const size_t xs = (size_t)60 * c->cfg.V * c->cfg.C_in, ts = 90 * 256;
    if (cnt_nm && (!cnt_mean || !cnt_std)) return fail(c, MOCHA_ERR_ARG, "mocha_encode: cnt_nm needs cnt_mean and cnt_std");
    const bool zn = cnt_nm != nullptr;                      // cnt itself is optional: the z-scored copy alone is a valid request
    return for_chunks(c, B, (hipStream_t)stream, [&](int b0, int b, hipStream_t s) -> int {
        int r = run_embed(c, X + b0 * xs, b, WS(c, "x5"), true, s);      // x5 is free again once the body block has read it
        if (r) return r;
        if ((r = run_encoder(c, WS(c, "x5"), b, encoded + b0 * ts, s))) return r;
        if (cnt || zn) {
            const InormExtra iex0 = IEX(c);
            LAUNCH(c, s, "mocha_instnorm", "mvn", 0.0, b * 90.0 * 256 * 4 * (1 + (cnt ? 1 : 0) + (zn ? 1 : 0)),
                   launch_instnorm(encoded + b0 * ts, cnt ? cnt + b0 * ts : nullptr, nullptr, zn ? cnt_mean : nullptr, zn ? cnt_std : nullptr,
                                   zn ? cnt_nm + b0 * ts : nullptr, b, 90, s, &iex0));
        }
        return 0;
    });
}

int mocha_decoder(mocha_ctx* c, const float* src_enc, const float* cha_enc, int B, float* out, void* stream) {
    int rc = ready(c, B); if (rc) return rc;
    NEED_PTRS(c, B, "mocha_decoder", src_enc, cha_enc, out);
    const size_t ts = 90 * 256;
    for (int i = 0; i < mocha_ctx::MAX_SETS; ++i) c->amax_enc_of[i] = c->amax_dec_of[i] = c->amax_tok_of[i] = nullptr;     // external activations: no carried bounds
    return for_chunks(c, B, (hipStream_t)stream, [&](int b0, int b, hipStream_t s) -> int {
        return run_decoder(c, src_enc + b0 * ts, cha_enc + b0 * ts, b, out + b0 * ts, s);
    });
}

int mocha_style_constants(mocha_ctx* c, const float* cha_enc, int B, float* gb, void* stream) {
    int rc = ready(c, B); if (rc) return rc;
    NEED_PTRS(c, B, "mocha_style_constants", cha_enc, gb);
    const size_t T = 90 * 256, G = (size_t)512 * c->cfg.dec_depth;
    return for_chunks(c, B, (hipStream_t)stream, [&](int b0, int b, hipStream_t s) -> int {
        InormExtra ex = IEX(c);
        if (c->style_f64) ex.mean64 = reinterpret_cast<double*>(WS(c, "smean64"));
        LAUNCH(c, s, "mocha_instnorm", "dec.in_cha", 0.0, b * 90.0 * 256 * 4 * 2,
               launch_instnorm(cha_enc + b0 * T, WS(c, "kin"), WS(c, "smean"), nullptr, nullptr, nullptr, b, 90, s, &ex));
        int r = run_style(c, b, s);
        if (r) return r;
        HIPCHK(c, hipMemcpyAsync(gb + b0 * G, WS(c, "gb"), (size_t)b * G * sizeof(float), hipMemcpyDeviceToDevice, s));
        return 0;
    });
}

int mocha_to_mot(mocha_ctx* c, const float* tokens, int B, float* Y, void* stream) {
    int rc = ready(c, B); if (rc) return rc;
    NEED_PTRS(c, B, "mocha_to_mot", tokens, Y);
    const size_t ts = 90 * 256, ys = (size_t)60 * c->cfg.V * c->cfg.C_in;
    for (int i = 0; i < mocha_ctx::MAX_SETS; ++i) c->amax_enc_of[i] = c->amax_dec_of[i] = c->amax_tok_of[i] = nullptr;     // external activations: no carried bounds
    return for_chunks(c, B, (hipStream_t)stream, [&](int b0, int b, hipStream_t s) -> int {
        return run_to_mot(c, tokens + b0 * ts, b, Y + b0 * ys, s);
    });
}

int mocha_forward(mocha_ctx* c, const float* src_X, const float* cha_X, int B, float* Y, void* stream) {
    int rc = ready(c, B); if (rc) return rc;
    NEED_PTRS(c, B, "mocha_forward", src_X, cha_X, Y);
    const size_t xs = (size_t)60 * c->cfg.V * c->cfg.C_in;
    return for_chunks(c, B, (hipStream_t)stream, [&](int b0, int b, hipStream_t s) -> int {
        int r;
        if ((r = run_embed(c, src_X + b0 * xs, b, WS(c, "x5"), true, s))) return r;
        if ((r = run_encoder(c, WS(c, "x5"), b, WS(c, "enc_s"), s))) return r;
        if ((r = run_embed(c, cha_X + b0 * xs, b, WS(c, "x5"), true, s))) return r;
        if ((r = run_encoder(c, WS(c, "x5"), b, WS(c, "enc_c"), s))) return r;
        if ((r = run_decoder(c, WS(c, "enc_s"), WS(c, "enc_c"), b, WS(c, "dec"), s))) return r;
        return run_to_mot(c, WS(c, "dec"), b, Y + b0 * xs, s);
    });
}

int mocha_forward_features(mocha_ctx* c, const float* src_X, const float* cha_X, int B, float* src_enc, float* cha_enc,
                           float* src_cnt, float* cha_cnt, void* stream) {
    int rc = mocha_encode(c, src_X, B, src_enc, src_cnt, nullptr, nullptr, nullptr, stream);
    if (rc) return rc;
    return mocha_encode(c, cha_X, B, cha_enc, cha_cnt, nullptr, nullptr, nullptr, stream);
}

// `current`: the bank becomes the context's current bank (mocha_bank_set); false for the transient bank of
// mocha_characterize_pair, which is swapped out again before the call returns and must not invalidate captured graphs
static int bank_set_impl(mocha_ctx* c, const float* cnt_nm, const float* encoded, int64_t N, int flags, void* stream, bool current) {
    int rc = ready(c, 0); if (rc) return rc;
    if (!cnt_nm || !encoded || N < 1 || N > (int64_t)1 << 30) return fail(c, MOCHA_ERR_ARG, "bad bank arguments");
    hipStream_t s = (hipStream_t)stream;
    const size_t D = 90 * 256;
    if (flags & MOCHA_BANK_BORROW) {
        c->bank_cnt = cnt_nm; c->bank_enc = encoded;
    } else {
        if (c->bank_cap < (size_t)N) {
            if (c->bank_cnt == c->bank_cnt_own || c->bank_enc == c->bank_enc_own) { c->bank_cnt = nullptr; c->bank_enc = nullptr; c->bank_N = 0; c->generation++; }
            c->bank_cap = 0;
            for (float** p : {&c->bank_cnt_own, &c->bank_enc_own})
                if (*p) { dev_free(c, *p); *p = nullptr; }
            if ((rc = dev_alloc(c, &c->bank_cnt_own, (size_t)N * D))) return rc;
            if ((rc = dev_alloc(c, &c->bank_enc_own, (size_t)N * D))) return rc;
            c->bank_cap = (size_t)N;
        }
        HIPCHK(c, hipMemcpyAsync(c->bank_cnt_own, cnt_nm, (size_t)N * D * sizeof(float), hipMemcpyDeviceToDevice, s));
        HIPCHK(c, hipMemcpyAsync(c->bank_enc_own, encoded, (size_t)N * D * sizeof(float), hipMemcpyDeviceToDevice, s));
        c->bank_cnt = c->bank_cnt_own; c->bank_enc = c->bank_enc_own;
    }
    if (c->bank_norm_cap < (size_t)N) {
        if (c->bank_norm) { dev_free(c, c->bank_norm); c->bank_norm = nullptr; }
        if ((rc = dev_alloc(c, &c->bank_norm, (size_t)N))) return rc;
        c->bank_norm_cap = (size_t)N;
    }
    c->bank_N = N;
    c->bank_is_bf16 = (flags & MOCHA_BANK_BF16) != 0;
    c->bank_tiled_valid = false; c->bank_tile32_valid = false;
    if (current) c->generation++;                         // a captured step has the previous bank's pointers and row count baked in
    // match scratch for every query count the workspace admits: a later match never allocates (capture-safe)
    for (int set = 0; set < mocha_ctx::MAX_SETS; ++set)
        if ((set == 0 || !c->wss[set].empty()) && (rc = ensure_match_scratch(c, set, set_windows(c, set), N, true))) return rc;
    // centroid of the bank: the many-query GEMM and the bf16 copy work on b - centroid (see do_match)
    if (!c->bank_center && (rc = dev_alloc(c, &c->bank_center, D))) return rc;
    if (!c->center_scratch && (rc = dev_alloc(c, &c->center_scratch, 2 * column_mean_scratch_doubles((int)D)))) return rc;
    LAUNCH(c, s, "mocha_column_mean", "bank.center", 0.0, 4.0 * N * D,
           launch_column_mean(c->bank_cnt, N, (int)D, c->bank_center, reinterpret_cast<double*>(c->center_scratch), s));
    if (c->bank_is_bf16) {
        if (c->bank_bf16_cap < (size_t)N) {
            if (c->bank_bf16) (void)hipFree(c->bank_bf16);
            c->bank_bf16 = nullptr;
            HIPCHK(c, hipMalloc(&c->bank_bf16, (size_t)N * D * 2));
            c->bank_bf16_cap = (size_t)N;
        }
        LAUNCH(c, s, "mocha_to_bf16", "bank.to_bf16", 0.0, 6.0 * N * D, launch_to_bf16(c->bank_cnt, c->bank_center, (int)D, c->bank_bf16, (int64_t)N * D, s));
        LAUNCH(c, s, "mocha_rownorm2_bf16", "bank.norms", 2.0 * N * D, 2.0 * N * D, launch_rownorm2_bf16(c->bank_bf16, c->bank_norm, N, (int)D, s));
    } else {
        LAUNCH(c, s, "mocha_rownorm2", "bank.norms", 2.0 * N * D, 4.0 * N * D, launch_rownorm2(c->bank_cnt, c->bank_center, c->bank_norm, N, (int)D, s));
    }
    // two-plane fp16 engine: the decoder's attention rows are bounded by the matched entries' largest magnitude, taken once here
    if (current) {
        c->amax_bank_ok = false;
        if (c->gemm_h2) {
            if (c->amax_bank_cap < (size_t)N) {
                HIPCHK(c, hipDeviceSynchronize());
                if (c->amax_bank) { dev_free(c, c->amax_bank); c->amax_bank = nullptr; c->amax_bank_cap = 0; }
                if ((rc = dev_alloc(c, &c->amax_bank, (size_t)N))) return rc;
                c->amax_bank_cap = (size_t)N;
            }
            HIPCHK(c, hipMemsetAsync(c->amax_bank, 0, (size_t)N * sizeof(float), s));
            for (int64_t n0 = 0; n0 < N; n0 += 65535)      // (the kernel's grid.y)
                LAUNCH(c, s, "mocha_absmax", "bank.absmax", 0.0, 4.0 * std::min<int64_t>(65535, N - n0) * D,
                       launch_absmax(c->bank_enc + (size_t)n0 * D, std::min<int64_t>(65535, N - n0), (long long)D, c->amax_bank + n0, s));
            c->amax_bank_ok = true;
        }
    }
    // few-query matching against a large fp32 bank scans its centred bf16 copy (half the bytes) and re-ranks exactly
    if (current) c->bank16f_valid = false;
    if (current && !c->bank_is_bf16 && c->scan16 && N >= SCAN16_MIN) {
        if (c->bank16f_cap < (size_t)N) {
            if (c->bank16f) (void)hipFree(c->bank16f);
            c->bank16f = nullptr; c->bank16f_cap = 0;
            HIPCHK(c, hipMalloc(&c->bank16f, (size_t)N * D * 2));
            c->bank16f_cap = (size_t)N;
        }
        if (c->bank_rho_cap < (size_t)N) {
            if (c->bank_rho) { dev_free(c, c->bank_rho); c->bank_rho = nullptr; }
            if ((rc = dev_alloc(c, &c->bank_rho, (size_t)N))) return rc;
            c->bank_rho_cap = (size_t)N;
        }
        LAUNCH(c, s, "mocha_to_bf16", "bank.to_bf16", 0.0, 6.0 * N * D, launch_to_bf16(c->bank_cnt, c->bank_center, (int)D, c->bank16f, (int64_t)N * D, s));
        LAUNCH(c, s, "mocha_rowresid", "bank.resid", 3.0 * N * D, 6.0 * N * D, launch_rowresid(c->bank_cnt, c->bank_center, c->bank16f, c->bank_rho, N, (int)D, s));
        c->bank16f_valid = true;
    }
    if (current) c->bank8_valid = false;
    if (current && c->bank16f_valid && c->scan8) {
        if (c->bank8_cap < (size_t)N) {
            HIPCHK(c, hipStreamSynchronize(s));                 // (a call still reading the old image)
            if (c->bank8) HIPCHK(c, hipFree(c->bank8));
            c->bank8 = nullptr; c->bank8_cap = 0;
            if (c->bank8_scale) { dev_free(c, c->bank8_scale); c->bank8_scale = nullptr; }
            if (c->bank8_rho) { dev_free(c, c->bank8_rho); c->bank8_rho = nullptr; }
            HIPCHK(c, hipMalloc(&c->bank8, (size_t)N * D));
            if ((rc = dev_alloc(c, &c->bank8_scale, (size_t)N))) return rc;
            if ((rc = dev_alloc(c, &c->bank8_rho, (size_t)N))) return rc;
            c->bank8_cap = (size_t)N;
        }
        LAUNCH(c, s, "mocha_to_i8", "bank.to_i8", 4.0 * N * D, 9.0 * N * D, launch_to_i8(c->bank_cnt, c->bank_center, c->bank8, c->bank8_scale, c->bank8_rho, N, (int)D, s));
        // a new bank: the byte stage gets another chance (its mode words, in every workspace set's scratch that exists)
        for (int st = 0; st < mocha_ctx::MAX_SETS; ++st)
            if (c->scan_keys[st]) HIPCHK(c, hipMemsetAsync(c->scan_keys[st] + match_scan8_mode_word(), 0, sizeof(unsigned long long), s));
        c->bank8_valid = true;
    }
    // decoder constants of every entry (round 5): IN(entry) and its AdaIN gamma / beta, read in place through frame_index by the decoder
    if (current) {
        c->bank_dec_valid = false;
        if (dec_cache_ok(c)) {
            const int L = c->cfg.dec_depth;
            if (c->bank_dec_cap < (size_t)N) {
                for (float** p : {&c->bank_kin, &c->bank_gb})
                    if (*p) { dev_free(c, *p); *p = nullptr; }
                c->bank_dec_cap = 0;
                // + N x (92 KB + 2 KB per decoder layer): about +50 % of the bank's footprint.  If the memory is not there the bank is set all
                // the same and the decoder computes its constants per call (the round-4 flow), instead of failing mocha_bank_set
                void* pk = nullptr; void* pg = nullptr;
                if (hipMalloc(&pk, (size_t)N * D * sizeof(float)) == hipSuccess && hipMalloc(&pg, (size_t)N * 512 * L * sizeof(float)) == hipSuccess) {
                    c->bank_kin = (float*)pk; c->bank_gb = (float*)pg;
                    c->owned.push_back(c->bank_kin); c->owned.push_back(c->bank_gb);
                    c->bank_dec_cap = (size_t)N;
                } else {
                    (void)hipGetLastError();                   // the failed allocation's sticky error
                    if (pk) (void)hipFree(pk);
                    if (pg) (void)hipFree(pg);
                }
            }
            if (c->bank_dec_cap >= (size_t)N) {
            const size_t rows = (size_t)std::min<int64_t>(N, 4096);
            if (c->style_scratch_rows < rows) {
                if (c->style_scratch) dev_free(c, reinterpret_cast<float*>(c->style_scratch));
                c->style_scratch = nullptr; c->style_scratch_rows = 0;
                float* f = nullptr;
                if ((rc = dev_alloc(c, &f, 2 * rows * (256 + (size_t)512 * L)))) return rc;
                c->style_scratch = reinterpret_cast<double*>(f); c->style_scratch_rows = rows;
            }
            if ((rc = build_dec_consts(c, c->bank_enc, N, c->bank_kin, c->bank_gb, c->style_scratch, c->style_scratch + rows * 256, (int64_t)rows, s))) return rc;
            c->bank_dec_valid = true;
            }
        }
    }
    // many-query matching against a small fp32 bank runs on the plane engine: centred bank as its packed image (6 B per value)
    c->bank_x3_valid = false;
    if (!c->bank_is_bf16 && c->gemm_x3 && N <= X3_BANK_MAX) {
        const size_t need = gemm_x3_packed_elems((int)N, (int)D);
        if (c->bank_x3_cap < need) {
            if (c->bank_x3) (void)hipFree(c->bank_x3);
            c->bank_x3 = nullptr; c->bank_x3_cap = 0;
            HIPCHK(c, hipMalloc((void**)&c->bank_x3, need * sizeof(unsigned short)));
            c->bank_x3_cap = need;
        }
        LAUNCH(c, s, "mocha_pack_x3", "bank.pack_x3", 0.0, 10.0 * N * D, launch_pack_x3(c->bank_cnt, (int)N, (int)D, c->bank_x3, s, c->bank_center));
        c->bank_x3_valid = true;
    }
    return 0;
}

int mocha_bank_set(mocha_ctx* c, const float* cnt_nm, const float* encoded, int64_t N, int flags, void* stream) {
    return bank_set_impl(c, cnt_nm, encoded, N, flags, stream, true);
}

int mocha_match(mocha_ctx* c, const float* query_nm, int Q, int32_t* idx, float* dist, void* stream) {
    int rc = ready(c, 0); if (rc) return rc;
    if (Q == 0) return 0;
    if (!query_nm || !idx || Q < 0) return fail(c, MOCHA_ERR_ARG, "bad match arguments");
    if (Q == 0) return 0;
    return do_match(c, query_nm, Q, idx, dist, (hipStream_t)stream);
}

int mocha_match_topk(mocha_ctx* c, const float* query_nm, int Q, int k, int32_t* idx, float* dist, void* stream) {
    int rc = ready(c, 0); if (rc) return rc;
    if (Q == 0) return 0;
    if (!query_nm || !idx || Q < 0 || k < 1 || k > 64) return fail(c, MOCHA_ERR_ARG, "bad top-k arguments (1 <= k <= 64)");
    if (!c->bank_cnt || c->bank_N <= 0) return fail(c, MOCHA_ERR_STATE, "no bank: call mocha_bank_set first");
    hipStream_t s = (hipStream_t)stream;
    const int D = 90 * 256;
    const int64_t N = c->bank_N;
    const size_t need = (size_t)8 * N;
    if (c->topk_keys_n < need) {                            // not on the per-frame path: allocated on first use
        HIPCHK(c, hipDeviceSynchronize());
        if (c->topk_keys) (void)hipFree(c->topk_keys);
        c->topk_keys = nullptr; c->topk_keys_n = 0;
        void* kp = nullptr;
        HIPCHK(c, hipMalloc(&kp, need * sizeof(unsigned long long)));
        c->topk_keys = (unsigned long long*)kp; c->topk_keys_n = need;
        c->generation++;
    }
    const float* q = query_nm;
    if (c->bank_is_bf16) {                                  // the bf16 bank holds bf16(b - centroid): centred queries
        if ((rc = grow(c, c->match_qc[0], (size_t)std::max(Q, 8) * D))) return rc;
        LAUNCH(c, s, "mocha_sub_rows", "match.center", 0.0, 8.0 * Q * D, launch_sub_rows(query_nm, c->bank_center, c->match_qc[0].p, Q, D, s));
        q = c->match_qc[0].p;
    }
    const void* bank = c->bank_is_bf16 ? (const void*)c->bank_bf16 : (const void*)c->bank_cnt;
    LAUNCH(c, s, "mocha_match_topk", "match.topk", 3.0 * Q * N * D, ((Q + 7) / 8) * (double)N * D * (c->bank_is_bf16 ? 2.0 : 4.0),
           launch_match_topk(bank, c->bank_is_bf16 ? 1 : 0, q, Q, N, D, k, c->topk_keys, idx, dist, s));
    return 0;
}

int mocha_bank_gather_blend(mocha_ctx* c, const int32_t* idx, const float* dist, float temperature, int Q, int k, float* out, void* stream) {
    int rc = ready(c, 0); if (rc) return rc;
    if (!c->bank_enc) return fail(c, MOCHA_ERR_STATE, "no bank: call mocha_bank_set first");
    if (Q == 0) return 0;
    if (!idx || !dist || !out || Q < 0 || k < 1 || k > 64 || !(temperature > 0.f)) return fail(c, MOCHA_ERR_ARG, "bad gather_blend arguments");
    hipStream_t s = (hipStream_t)stream;
    LAUNCH(c, s, "mocha_gather_blend", "bank.gather_blend", 2.0 * Q * k * 90 * 256, Q * (k + 1.0) * 90 * 256 * 4,
           launch_gather_blend(c->bank_enc, idx, dist, temperature, out, Q, k, 90 * 256, c->bank_N, s));
    return 0;
}

int mocha_bank_gather(mocha_ctx* c, const int32_t* idx, int Q, float* out, void* stream) {
    int rc = ready(c, 0); if (rc) return rc;
    if (!c->bank_enc) return fail(c, MOCHA_ERR_STATE, "no bank: call mocha_bank_set first");
    if (Q == 0) return 0;
    if (!idx || !out || Q < 0) return fail(c, MOCHA_ERR_ARG, "bad gather arguments");
    hipStream_t s = (hipStream_t)stream;
    LAUNCH(c, s, "mocha_gather_rows", "bank.gather", 0.0, Q * 90.0 * 256 * 8, launch_gather_rows(c->bank_enc, idx, out, Q, 90 * 256, c->bank_N, s));
    return 0;
}

static int characterize_impl(mocha_ctx* c, const float* src_X, int B, const float* cnt_mean, const float* cnt_std, float* Y,
                             int32_t* idx, void* stream, bool raw) {
    int rc = ready(c, B); if (rc) return rc;
    if (!c->bank_cnt) return fail(c, MOCHA_ERR_STATE, "no bank: call mocha_bank_set first");
    if (B == 0) return 0;
    if (!cnt_mean || !cnt_std || !src_X || !Y) return fail(c, MOCHA_ERR_ARG, "null argument");
    const size_t ys = (size_t)60 * c->cfg.V * c->cfg.C_in;
    const size_t xs = raw ? (size_t)60 * (c->cfg.V + 1) * c->cfg.C_in : ys;
    return for_chunks(c, B, (hipStream_t)stream, [&](int b0, int b, hipStream_t s) -> int {
        int r;
        int32_t* ix = idx ? idx + b0 : c->idx_ws[c->cur];
        if ((r = run_embed(c, src_X + b0 * xs, b, WS(c, "x5"), true, s, raw))) return r;
        if ((r = run_encoder(c, WS(c, "x5"), b, WS(c, "enc_s"), s))) return r;
        // cnt, its z-score and - the bank's centroid is known - the matcher's centred queries in one pass
        // (as one bf16 plane where the many-query pass against a bf16 bank will read them: no mocha_center_bf16 launch)
        const bool q16 = c->bank_is_bf16 && b > 8;
        const bool sel2 = use_select2(c, b);                     // the selection's row statistics (and the second bf16 plane) come from this pass too
        if (sel2 && (r = ensure_match_scratch(c, c->cur, b, c->bank_N, false))) return r;
        InormExtra ex = IEX(c); ex.centre = c->bank_center;
        if (q16) ex.zc16 = reinterpret_cast<unsigned short*>(WS(c, "qc")); else ex.zc = WS(c, "qc");
        if (sel2) { ex.qstat = c->match_qstat[c->cur].p; if (q16 && c->match_planes == 2) ex.plane_stride = (long long)b * 90 * 256; }
        LAUNCH(c, s, "mocha_instnorm", "mvn", 0.0, b * 90.0 * 256 * 4 * (q16 ? (sel2 ? 3.0 : 2.5) : 3.0), launch_instnorm(WS(c, "enc_s"), nullptr, nullptr, cnt_mean, cnt_std, WS(c, "qnm"), b, 90, s, &ex));
        if ((r = do_match(c, WS(c, "qnm"), b, ix, nullptr, s, WS(c, "qc"), q16, sel2))) return r;
        // decoder on cha_encoded[frame_index]: its first kernel gathers the rows itself
        if ((r = run_decoder(c, WS(c, "enc_s"), nullptr, b, WS(c, "dec"), s, c->bank_enc, ix, c->bank_N,
                             c->bank_dec_valid ? c->bank_kin : nullptr, c->bank_dec_valid ? c->bank_gb : nullptr))) return r;
        return run_to_mot(c, WS(c, "dec"), b, Y + b0 * ys, s, raw);
    });
}

int mocha_characterize(mocha_ctx* c, const float* src_X, int B, const float* cnt_mean, const float* cnt_std, float* Y,
                       int32_t* idx, void* stream) {
    return characterize_impl(c, src_X, B, cnt_mean, cnt_std, Y, idx, stream, false);
}

int mocha_characterize_raw(mocha_ctx* c, const float* src_X_raw, int B, const float* cnt_mean, const float* cnt_std, float* Y,
                           int32_t* idx, void* stream) {
    return characterize_impl(c, src_X_raw, B, cnt_mean, cnt_std, Y, idx, stream, true);
}

// Demo pair in one pass (test_fullframework.py:188-194 for both clips, then :293-296, 438-443, 465-467): the character clip
// becomes the bank, the source clip is characterized against it.  Both clips go through mot_embedding / encoder / cnt in the
// same launches (twice the tiles per launch: less tile rounding, fewer prologue / epilogue phases); the bank lives in the
// workspace for the duration of the call and the context's own bank (mocha_bank_set) is left untouched.
static int characterize_pair_impl(mocha_ctx* c, const float* src_X, int B_src, const float* cha_X, int B_cha, const float* cnt_mean,
                                  const float* cnt_std, float* Y, int32_t* idx, float* cha_encoded, float* cha_cnt_nm, void* stream,
                                  bool raw) {
    if (!c) return MOCHA_ERR_ARG;
    if (B_src < 0 || B_cha < 1) return fail(c, MOCHA_ERR_ARG, "characterize_pair: needs B_src >= 0 and B_cha >= 1");
    const long long total = (long long)B_src + B_cha;
    if (total > c->max_chunk)
        return fail(c, MOCHA_ERR_ARG, "characterize_pair: %lld windows exceed the workspace limit of %d (raise it with mocha_reserve, or "
                    "use mocha_encode + mocha_bank_set + mocha_characterize)", total, c->max_chunk);
    int rc = ready(c, (int)total); if (rc) return rc;
    if (!cnt_mean || !cnt_std || !cha_X || (B_src > 0 && (!src_X || !Y))) return fail(c, MOCHA_ERR_ARG, "null argument");
    hipStream_t s = (hipStream_t)stream;
    c->cur = 0;
    const size_t T = 90 * 256;
    const int B = (int)total;
    // character windows first: rows [0, B_cha) of every buffer are the bank, rows [B_cha, B) the queries
    if ((rc = run_embed(c, cha_X, B_cha, WS(c, "x5"), true, s, raw, src_X, B_src))) return rc;
    if ((rc = run_encoder(c, WS(c, "x5"), B, WS(c, "enc_s"), s))) return rc;
    // The transient bank's decoder constants (IN(entry), gamma / beta of the character windows) depend on the encoder's output alone and
    // are needed by the decoder only: with "pair_overlap" they run on the context's internal stream BESIDE the matching chain (cnt, centroid,
    // norms, plane image, coarse GEMM, selection - small latency-bound launches), forked / joined with events (capture-safe, as for_chunks).
    const float *kin_t = nullptr, *gb_t = nullptr;
    bool forked = false;
    if (B_src > 0 && dec_cache_ok(c)) {
        hipStream_t cs = s;
        if (c->pair_overlap && c->aux) {
            HIPCHK(c, hipEventRecord(c->ev_fork, s));
            HIPCHK(c, hipStreamWaitEvent(c->aux, c->ev_fork, 0));
            cs = c->aux; forked = true;
        }
        rc = build_dec_consts(c, WS(c, "enc_s"), B_cha, WS(c, "kin"), WS(c, "gb"), reinterpret_cast<double*>(WS(c, "smean64")),
                              reinterpret_cast<double*>(WS(c, "s1d")), B_cha, cs);
        if (forked) {                                     // join even when a launch failed: the caller's stream must not be left forked
            const hipError_t ej = hipEventRecord(c->ev_join, c->aux);
            if (!rc && ej != hipSuccess) rc = fail(c, MOCHA_ERR_HIP, "hipEventRecord: %s", hipGetErrorString(ej));
        }
        if (rc) { if (forked) (void)hipStreamWaitEvent(s, c->ev_join, 0); return rc; }
        kin_t = WS(c, "kin"); gb_t = WS(c, "gb");
    }
    auto join = [&]() -> int { if (forked) { forked = false; HIPCHK(c, hipStreamWaitEvent(s, c->ev_join, 0)); } return 0; };
    // everything between the fork and the join runs in a lambda: an early error return must not skip the join
    rc = [&]() -> int {
        const InormExtra iex0 = IEX(c);
        LAUNCH(c, s, "mocha_instnorm", "mvn", 0.0, B * 90.0 * 256 * 4 * 2,
               launch_instnorm(WS(c, "enc_s"), nullptr, nullptr, cnt_mean, cnt_std, WS(c, "qnm"), B, 90, s, &iex0));
        if (cha_encoded) HIPCHK(c, hipMemcpyAsync(cha_encoded, WS(c, "enc_s"), (size_t)B_cha * T * sizeof(float), hipMemcpyDeviceToDevice, s));
        if (cha_cnt_nm) HIPCHK(c, hipMemcpyAsync(cha_cnt_nm, WS(c, "qnm"), (size_t)B_cha * T * sizeof(float), hipMemcpyDeviceToDevice, s));
        return 0;
    }();
    if (rc) { (void)join(); return rc; }
    if (B_src == 0) return 0;
    // transient bank: swap the context's bank state out, borrow the workspace rows, restore afterwards
    // (everything bank_set_impl derives from a bank has a pair_* twin: norms, centroid, packed plane image - the user's
    // current bank and what was derived from it are exactly as before when the call returns)
    struct Saved { const float *cnt, *enc; int64_t N; bool bf16; float* norm; size_t norm_cap; float* center;
                   unsigned short* x3; size_t x3_cap; bool x3_valid; bool v16; } sv{
        c->bank_cnt, c->bank_enc, c->bank_N, c->bank_is_bf16, c->bank_norm, c->bank_norm_cap, c->bank_center,
        c->bank_x3, c->bank_x3_cap, c->bank_x3_valid, c->bank16f_valid};
    c->bank16f_valid = false;                             // the bf16 copy belongs to the user's bank, not to the transient one
    c->bank_norm = c->pair_norm; c->bank_norm_cap = c->pair_norm_cap; c->bank_center = c->pair_center;
    c->bank_x3 = c->pair_x3; c->bank_x3_cap = c->pair_x3_cap; c->bank_x3_valid = false;
    rc = bank_set_impl(c, WS(c, "qnm"), WS(c, "enc_s"), B_cha, MOCHA_BANK_BORROW, stream, false);
    int32_t* ix = idx ? idx : c->idx_ws[0];
    if (!rc) rc = do_match(c, WS(c, "qnm") + (size_t)B_cha * T, B_src, ix, nullptr, s);
    c->pair_norm = c->bank_norm; c->pair_norm_cap = c->bank_norm_cap; c->pair_center = c->bank_center;
    c->pair_x3 = c->bank_x3; c->pair_x3_cap = c->bank_x3_cap;
    c->bank_x3 = sv.x3; c->bank_x3_cap = sv.x3_cap; c->bank_x3_valid = sv.x3_valid; c->bank16f_valid = sv.v16;
    c->bank_cnt = sv.cnt; c->bank_enc = sv.enc; c->bank_N = sv.N; c->bank_is_bf16 = sv.bf16; c->bank_norm = sv.norm; c->bank_norm_cap = sv.norm_cap;
    c->bank_center = sv.center;
    { const int rj = join(); if (!rc) rc = rj; }          // the decoder constants are complete (the caller's stream is never left forked)
    if (rc) return rc;
    // decoder on the matched character rows: the transient bank's encoded rows are rows [0, B_cha) of the workspace, their decoder
    // constants (as mocha_bank_set makes them for a user's bank) rows [0, B_cha) of "kin" / "gb"; read in place through the indices
    if ((rc = run_decoder(c, WS(c, "enc_s") + (size_t)B_cha * T, nullptr, B_src, WS(c, "dec"), s, WS(c, "enc_s"), ix, B_cha, kin_t, gb_t))) return rc;
    return run_to_mot(c, WS(c, "dec"), B_src, Y, s, raw);
}

int mocha_characterize_pair(mocha_ctx* c, const float* src_X, int B_src, const float* cha_X, int B_cha, const float* cnt_mean,
                            const float* cnt_std, float* Y, int32_t* idx, float* cha_encoded, float* cha_cnt_nm, void* stream) {
    return characterize_pair_impl(c, src_X, B_src, cha_X, B_cha, cnt_mean, cnt_std, Y, idx, cha_encoded, cha_cnt_nm, stream, false);
}

int mocha_characterize_pair_raw(mocha_ctx* c, const float* src_X_raw, int B_src, const float* cha_X_raw, int B_cha, const float* cnt_mean,
                                const float* cnt_std, float* Y, int32_t* idx, float* cha_encoded, float* cha_cnt_nm, void* stream) {
    return characterize_pair_impl(c, src_X_raw, B_src, cha_X_raw, B_cha, cnt_mean, cnt_std, Y, idx, cha_encoded, cha_cnt_nm, stream, true);
}

// ------------------------------------------------------------------------------------------- captured per-window step
// BASELINE configs[4]: a clip streamed one 60-frame window per step.  The whole step (mot_embedding, +pos_emb, encoder, cnt,
// z-score, bank scan, gather, decoder, to_mot: test_fullframework.py:438-443, 465-467) is captured once into a HIP graph and
// replayed; the graph is keyed on the buffer pointers and on the context generation, and re-captured when either changes.
int mocha_step_graph_lane(mocha_ctx* c, int lane, const float* X1, const float* cnt_mean, const float* cnt_std, float* Y1, int32_t* idx,
                          int raw, void* stream) {
    if (!c) return MOCHA_ERR_ARG;
    if (lane < 0 || lane >= c->lanes) return fail(c, MOCHA_ERR_ARG, "lane %d: the context has %d lane(s) (mocha_set_option \"lanes\")", lane, c->lanes);
    int rc = ready(c, 1); if (rc) return rc;
    if (!X1 || !cnt_mean || !cnt_std || !Y1 || !idx) return fail(c, MOCHA_ERR_ARG, "null argument");
    if (!c->bank_cnt) return fail(c, MOCHA_ERR_STATE, "no bank: call mocha_bank_set first");
    if (c->wss[lane].empty()) return fail(c, MOCHA_ERR_STATE, "lane %d has no workspace", lane);
    hipStream_t s = (hipStream_t)stream;
    auto& g = c->step[lane];
    const bool hit = g.exec && g.x == X1 && g.mean == cnt_mean && g.sd == cnt_std && g.y == Y1 && g.idx == idx &&
                     g.generation == c->generation && g.raw == (raw != 0);
    if (!hit) {
        if (c->prof_on) return fail(c, MOCHA_ERR_STATE, "mocha_step_graph: stop profiling before capturing");
        // make sure nothing inside the captured region allocates (scratch for one query against the current bank)
        if ((rc = ensure_match_scratch(c, lane, 8, c->bank_N, true))) return rc;
        if (!c->cap_stream) HIPCHK(c, hipStreamCreateWithFlags(&c->cap_stream, hipStreamNonBlocking));
        if (g.exec) { (void)hipGraphExecDestroy(g.exec); g.exec = nullptr; }
        if (g.graph) { (void)hipGraphDestroy(g.graph); g.graph = nullptr; }
        const int64_t gen0 = c->generation;
        HIPCHK(c, hipStreamBeginCapture(c->cap_stream, hipStreamCaptureModeRelaxed));
        c->lane = lane;                                   // the step's kernels use this lane's workspace set and match scratch
        rc = characterize_impl(c, X1, 1, cnt_mean, cnt_std, Y1, idx, c->cap_stream, raw != 0);
        c->lane = 0; c->cur = 0;
        hipGraph_t graph = nullptr;
        const hipError_t ee = hipStreamEndCapture(c->cap_stream, &graph);
        if (rc) { if (graph) (void)hipGraphDestroy(graph); return rc; }
        if (ee != hipSuccess) return fail(c, MOCHA_ERR_HIP, "hipStreamEndCapture: %s", hipGetErrorString(ee));
        if (c->generation != gen0) { (void)hipGraphDestroy(graph); return fail(c, MOCHA_ERR_STATE, "a buffer was replaced during capture"); }
        hipGraphExec_t exec = nullptr;
        const hipError_t ei = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
        if (ei != hipSuccess) { (void)hipGraphDestroy(graph); return fail(c, MOCHA_ERR_HIP, "hipGraphInstantiate: %s", hipGetErrorString(ei)); }
        g.exec = exec; g.graph = graph; g.x = X1; g.mean = cnt_mean; g.sd = cnt_std; g.y = Y1; g.idx = idx;
        g.generation = c->generation; g.raw = raw != 0;
    }
    HIPCHK(c, hipGraphLaunch(g.exec, s));
    return 0;
}

int mocha_step_graph(mocha_ctx* c, const float* X1, const float* cnt_mean, const float* cnt_std, float* Y1, int32_t* idx,
                     int raw, void* stream) {
    return mocha_step_graph_lane(c, 0, X1, cnt_mean, cnt_std, Y1, idx, raw, stream);
}

// ------------------------------------------------------------------------------------------- RCCL (multi-GPU set-up)
// One process per GPU; the only exchange on the path is the one-time broadcast of the character bank.
// RCCL is resolved at run time (dlopen of librccl.so.1) so that single-GPU users never load it.
namespace {
struct Rccl {
    void* h = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*CommUserRank)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*GetVersion)(int*) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
Rccl g_rccl;
std::string g_rccl_path;          // mocha_set_rccl_library: a process that already holds an RCCL (PyTorch ships its own) should use that one

int rccl_load(mocha_ctx* c) {
    if (g_rccl.h) return 0;
    void* h = nullptr;
    if (!g_rccl_path.empty()) {
        h = dlopen(g_rccl_path.c_str(), RTLD_NOW | RTLD_LOCAL);
        if (!h) return fail(c, MOCHA_ERR_STATE, "cannot load RCCL from '%s': %s", g_rccl_path.c_str(), dlerror());
    }
    if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
    if (!h) return fail(c, MOCHA_ERR_STATE, "cannot load RCCL (librccl.so.1): %s", dlerror());
    Rccl r; r.h = h;
    bool ok = true;
    auto sym = [&](const char* n) { void* p = dlsym(h, n); if (!p) ok = false; return p; };
    r.GetUniqueId = (decltype(r.GetUniqueId))sym("ncclGetUniqueId");
    r.CommInitRank = (decltype(r.CommInitRank))sym("ncclCommInitRank");
    r.CommDestroy = (decltype(r.CommDestroy))sym("ncclCommDestroy");
    r.CommCount = (decltype(r.CommCount))sym("ncclCommCount");
    r.CommUserRank = (decltype(r.CommUserRank))sym("ncclCommUserRank");
    r.Broadcast = (decltype(r.Broadcast))sym("ncclBroadcast");
    r.AllGather = (decltype(r.AllGather))sym("ncclAllGather");
    r.Send = (decltype(r.Send))sym("ncclSend");
    r.Recv = (decltype(r.Recv))sym("ncclRecv");
    r.GroupStart = (decltype(r.GroupStart))sym("ncclGroupStart");
    r.GroupEnd = (decltype(r.GroupEnd))sym("ncclGroupEnd");
    r.GetVersion = (decltype(r.GetVersion))sym("ncclGetVersion");
    r.GetErrorString = (decltype(r.GetErrorString))sym("ncclGetErrorString");
    if (!ok) { dlclose(h); return fail(c, MOCHA_ERR_STATE, "librccl lacks an expected entry point"); }
    g_rccl = r;
    return 0;
}
#define NCCLCHK(c, expr)                                                                                                  \
    do {                                                                                                                  \
        ncclResult_t r__ = (expr);                                                                                        \
        if (r__ != ncclSuccess) return fail((c), MOCHA_ERR_HIP, "%s failed: %s", #expr, g_rccl.GetErrorString(r__));    \
    } while (0)

// Broadcast `count` floats from `root` so that every xGMI link of the root carries traffic: the root sends chunk r to
// rank r (grouped point-to-point = scatter), then an in-place all-gather completes every rank's copy.  A flat
// ncclBroadcast is a ring through the root's neighbours and bound by one link (~153 GB/s); scatter + all-gather moves
// 1/world of the data per link and step.  The count % world tail (a few floats) goes through one small ncclBroadcast.
// the plan of one chunked broadcast: rank r owns floats [r * chunk, (r + 1) * chunk), the last `tail` floats are broadcast whole
struct BcastPlan { size_t chunk, tail, tail_off; };
BcastPlan bcast_plan(size_t count, int world) {
    BcastPlan p;
    p.chunk = count / (size_t)world;
    p.tail_off = p.chunk * (size_t)world;
    p.tail = count - p.tail_off;
    return p;
}

int bcast_chunked(mocha_ctx* c, ncclComm_t comm, int world, int rank, int root, float* buf, size_t count, hipStream_t s) {
    if (world == 1 || count == 0) return 0;
    const BcastPlan pl = bcast_plan(count, world);
    if (pl.chunk > 0) {
        // inside a group no call may return early: a group left open would swallow every later RCCL call of this thread.
        // The first error is kept, the group is always closed.
        ncclResult_t first = ncclSuccess;
        const char* what = "";
        auto keep = [&](ncclResult_t r, const char* w) { if (r != ncclSuccess && first == ncclSuccess) { first = r; what = w; } };
        NCCLCHK(c, g_rccl.GroupStart());
        if (rank == root) {
            for (int r = 0; r < world && first == ncclSuccess; ++r)
                if (r != root) keep(g_rccl.Send(buf + pl.chunk * (size_t)r, pl.chunk, ncclFloat32, r, comm, s), "ncclSend");
        } else {
            keep(g_rccl.Recv(buf + pl.chunk * (size_t)rank, pl.chunk, ncclFloat32, root, comm, s), "ncclRecv");
        }
        keep(g_rccl.GroupEnd(), "ncclGroupEnd");
        if (first != ncclSuccess) return fail(c, MOCHA_ERR_HIP, "%s failed in the bank scatter: %s", what, g_rccl.GetErrorString(first));
        NCCLCHK(c, g_rccl.AllGather(buf + pl.chunk * (size_t)rank, buf, pl.chunk, ncclFloat32, comm, s));
    }
    if (pl.tail > 0) NCCLCHK(c, g_rccl.Broadcast(buf + pl.tail_off, buf + pl.tail_off, pl.tail, ncclFloat32, root, comm, s));
    return 0;
}
}  // namespace

int mocha_bcast_plan(int64_t count, int world, int rank, int64_t out[4]) {
    if (count < 0 || world < 1 || rank < 0 || rank >= world || !out) return MOCHA_ERR_ARG;
    const BcastPlan p = bcast_plan((size_t)count, world);
    out[0] = (int64_t)(p.chunk * (size_t)rank); out[1] = (int64_t)p.chunk; out[2] = (int64_t)p.tail_off; out[3] = (int64_t)p.tail;
    return 0;
}

int mocha_set_rccl_library(const char* path) {
    if (g_rccl.h) return MOCHA_ERR_STATE;              // already resolved: too late to switch
    g_rccl_path = path ? path : "";
    return 0;
}

int mocha_comm_unique_id(mocha_ctx* c, void* id128) {
    if (!c || !id128) return fail(c, MOCHA_ERR_ARG, "null argument");
    int rc = rccl_load(c); if (rc) return rc;
    ncclUniqueId id;
    NCCLCHK(c, g_rccl.GetUniqueId(&id));
    memcpy(id128, id.internal, NCCL_UNIQUE_ID_BYTES);
    return 0;
}

int mocha_comm_init(mocha_ctx* c, const void* id128, int nranks, int rank) {
    if (!c || !id128 || nranks < 1 || rank < 0 || rank >= nranks) return fail(c, MOCHA_ERR_ARG, "bad communicator arguments");
    int rc = rccl_load(c); if (rc) return rc;
    if (c->comm) return fail(c, MOCHA_ERR_STATE, "communicator already initialised");
    HIPCHK(c, hipSetDevice(c->device));
    ncclUniqueId id;
    memcpy(id.internal, id128, NCCL_UNIQUE_ID_BYTES);
    NCCLCHK(c, g_rccl.CommInitRank(&c->comm, nranks, id, rank));
    c->comm_rank = rank; c->comm_size = nranks;
    return 0;
}

int mocha_comm_info(mocha_ctx* c, mocha_comm_info_t* out) {
    if (!c || !out) return fail(c, MOCHA_ERR_ARG, "null argument");
    if (!c->comm || !g_rccl.h) return fail(c, MOCHA_ERR_STATE, "no communicator: call mocha_comm_init first");
    memset(out, 0, sizeof *out);
    NCCLCHK(c, g_rccl.CommCount(c->comm, &out->nranks));
    NCCLCHK(c, g_rccl.CommUserRank(c->comm, &out->rank));
    NCCLCHK(c, g_rccl.GetVersion(&out->rccl_version));
    out->device = c->device;
    HIPCHK(c, hipDeviceGetPCIBusId(out->pci_bus_id, (int)sizeof out->pci_bus_id, c->device));
    Dl_info di;
    if (dladdr(reinterpret_cast<const void*>(g_rccl.CommInitRank), &di) && di.dli_fname) {
        char real[4096];
        const char* path = realpath(di.dli_fname, real) ? real : di.dli_fname;
        snprintf(out->library, sizeof out->library, "%s", path);
    }
    return 0;
}

int mocha_comm_destroy(mocha_ctx* c) {
    if (!c) return MOCHA_ERR_ARG;
    if (c->comm && g_rccl.h) { (void)g_rccl.CommDestroy(c->comm); }
    c->comm = nullptr; c->comm_rank = 0; c->comm_size = 1;
    return 0;
}

int mocha_bank_broadcast(mocha_ctx* c, void* comm_, int root, int64_t N, int flags, void* stream) {
    int rc = ready(c, 0); if (rc) return rc;
    if ((rc = rccl_load(c))) return rc;
    ncclComm_t comm = comm_ ? (ncclComm_t)comm_ : c->comm;
    if (!comm) return fail(c, MOCHA_ERR_STATE, "no communicator: call mocha_comm_init first or pass one");
    int world = 1, rank = 0;
    NCCLCHK(c, g_rccl.CommCount(comm, &world));
    NCCLCHK(c, g_rccl.CommUserRank(comm, &rank));
    if (root < 0 || root >= world || N < 1 || N > (int64_t)1 << 30) return fail(c, MOCHA_ERR_ARG, "bad bank_broadcast arguments");
    hipStream_t s = (hipStream_t)stream;
    const size_t D = 90 * 256;
    const bool want_bf16 = (flags & MOCHA_BANK_BF16) != 0;
    // Header first: EVERY rank contributes {entries, bf16?} as it understands the call - the root from the bank it is about to send
    // (-1 entries: it has none of the size the call names), the others from their arguments - the headers are all-gathered, and
    // every rank checks all of them before any payload moves.  All ranks see the same headers, so a disagreement fails loudly on
    // ALL ranks instead of leaving some of them inside a collective the others never enter.
    if (world > 4096) return fail(c, MOCHA_ERR_ARG, "bank_broadcast: %d ranks", world);
    if (!c->bcast_hdr) HIPCHK(c, hipMalloc((void**)&c->bcast_hdr, 2 * 4096 * sizeof(long long)));
    std::vector<long long> hdr((size_t)2 * world, 0);
    long long mine[2] = {(long long)N, want_bf16 ? 1 : 0};
    if (rank == root) { mine[0] = (c->bank_cnt && c->bank_N == N) ? (long long)N : -1; mine[1] = c->bank_is_bf16 ? 1 : 0; }
    HIPCHK(c, hipMemcpyAsync(c->bcast_hdr + 2 * rank, mine, sizeof mine, hipMemcpyHostToDevice, s));
    if (world > 1) NCCLCHK(c, g_rccl.AllGather(c->bcast_hdr + 2 * rank, c->bcast_hdr, 2, ncclInt64, comm, s));
    HIPCHK(c, hipMemcpyAsync(hdr.data(), c->bcast_hdr, hdr.size() * sizeof(long long), hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipStreamSynchronize(s));
    if (hdr[2 * root] != (long long)N || hdr[2 * root] < 0) {
        if (hdr[2 * root] < 0)
            return fail(c, MOCHA_ERR_STATE, "bank_broadcast: the root (rank %d) has no current bank of the number of entries it was called with", root);
        return fail(c, MOCHA_ERR_STATE, "bank_broadcast: the root (rank %d) has no current bank of %lld entries (it announced %lld)", root,
                    (long long)N, hdr[2 * root]);
    }
    for (int r = 0; r < world; ++r) {
        if (hdr[2 * r] != hdr[2 * root])
            return fail(c, MOCHA_ERR_ARG, "bank_broadcast: rank %d asks for %lld entries, the root (rank %d) sends %lld: every rank must name the same bank",
                        r, hdr[2 * r], root, hdr[2 * root]);
        if (hdr[2 * r + 1] != hdr[2 * root + 1])
            return fail(c, MOCHA_ERR_ARG, "bank_broadcast: the root's bank was set %s MOCHA_BANK_BF16 but rank %d asks for the opposite: every "
                        "rank must match against the same bank", hdr[2 * root + 1] ? "with" : "without", r);
    }
    if (rank != root) {
        if (c->bank_cap < (size_t)N) {
            HIPCHK(c, hipDeviceSynchronize());
            // the current bank may be the copy that is about to be freed: no dangling pointers if an allocation below fails
            if (c->bank_cnt == c->bank_cnt_own || c->bank_enc == c->bank_enc_own) { c->bank_cnt = nullptr; c->bank_enc = nullptr; c->bank_N = 0; c->generation++; }
            c->bank_cap = 0;
            for (float** p : {&c->bank_cnt_own, &c->bank_enc_own})
                if (*p) { dev_free(c, *p); *p = nullptr; }
            if ((rc = dev_alloc(c, &c->bank_cnt_own, (size_t)N * D))) return rc;
            if ((rc = dev_alloc(c, &c->bank_enc_own, (size_t)N * D))) return rc;
            c->bank_cap = (size_t)N;
        }
    }
    // the root's own chunks are rewritten in place with identical values by the all-gather (a borrowed bank stays unchanged)
    float* cnt = rank == root ? const_cast<float*>(c->bank_cnt) : c->bank_cnt_own;
    float* enc = rank == root ? const_cast<float*>(c->bank_enc) : c->bank_enc_own;
    if ((rc = bcast_chunked(c, comm, world, rank, root, cnt, (size_t)N * D, s))) return rc;
    if ((rc = bcast_chunked(c, comm, world, rank, root, enc, (size_t)N * D, s))) return rc;
    if (rank == root) return 0;
    // derived data (centroid, norms, bf16 copy) is recomputed locally: deterministic kernels, same result on every rank
    return bank_set_impl(c, c->bank_cnt_own, c->bank_enc_own, N, (flags & MOCHA_BANK_BF16) | MOCHA_BANK_BORROW, stream, true);
}

int mocha_set_pose_norm(mocha_ctx* c, const float* x_mean, const float* x_std, const float* y_mean, const float* y_std) {
    if (!c || !x_mean || !x_std || !y_mean || !y_std) return fail(c, MOCHA_ERR_ARG, "null argument");
    HIPCHK(c, hipSetDevice(c->device));
    const size_t nn = (size_t)(c->cfg.V + 1) * c->cfg.C_in;
    for (size_t i = 0; i < nn; ++i)
        if (!(x_std[i] != 0.f)) return fail(c, MOCHA_ERR_ARG, "x_std[%zu] is zero", i);
    if (!c->pose_norm) { int rc = dev_alloc(c, &c->pose_norm, 4 * nn); if (rc) return rc; }
    const float* src[4] = {x_mean, x_std, y_mean, y_std};
    for (int k = 0; k < 4; ++k) HIPCHK(c, hipMemcpy(c->pose_norm + k * nn, src[k], nn * sizeof(float), hipMemcpyHostToDevice));
    return 0;
}

int mocha_encode_raw(mocha_ctx* c, const float* X_raw, int B, float* encoded, float* cnt, const float* cnt_mean,
                     const float* cnt_std, float* cnt_nm, void* stream) {
    int rc = ready(c, B); if (rc) return rc;
    NEED_PTRS(c, B, "mocha_encode_raw", X_raw, encoded);
    const size_t xs = (size_t)60 * (c->cfg.V + 1) * c->cfg.C_in, ts = 90 * 256;
    if (cnt_nm && (!cnt_mean || !cnt_std)) return fail(c, MOCHA_ERR_ARG, "mocha_encode_raw: cnt_nm needs cnt_mean and cnt_std");
    const bool zn = cnt_nm != nullptr;
    return for_chunks(c, B, (hipStream_t)stream, [&](int b0, int b, hipStream_t s) -> int {
        int r;
        if ((r = run_embed(c, X_raw + b0 * xs, b, WS(c, "x5"), true, s, true))) return r;
        if ((r = run_encoder(c, WS(c, "x5"), b, encoded + b0 * ts, s))) return r;
        if (cnt || zn) {
            const InormExtra iex0 = IEX(c);
            LAUNCH(c, s, "mocha_instnorm", "mvn", 0.0, b * 90.0 * 256 * 4 * (1 + (cnt ? 1 : 0) + (zn ? 1 : 0)),
                   launch_instnorm(encoded + b0 * ts, cnt ? cnt + b0 * ts : nullptr, nullptr, zn ? cnt_mean : nullptr, zn ? cnt_std : nullptr,
                                   zn ? cnt_nm + b0 * ts : nullptr, b, 90, s, &iex0));
        }
        return 0;
    });
}

// ------------------------------------------------------------------------------------------- CVAE sampler
static void cvae_expectations(mocha_ctx* c) {
    if (!c->cvae_expect.empty()) return;
    auto& e = c->cvae_expect;
    const int64_t d = 256, ff = 512;
    e["prior_net.mu_token"] = {1, 1, d};
    e["prior_net.logvar_token"] = {1, 1, d};
    auto attn = [&](const std::string& p) {
        e[p + ".in_proj_weight"] = {3 * d, d}; e[p + ".in_proj_bias"] = {3 * d};
        e[p + ".out_proj.weight"] = {d, d}; e[p + ".out_proj.bias"] = {d};
    };
    auto ffn = [&](const std::string& p, int norms) {
        e[p + ".linear1.weight"] = {ff, d}; e[p + ".linear1.bias"] = {ff};
        e[p + ".linear2.weight"] = {d, ff}; e[p + ".linear2.bias"] = {d};
        for (int n = 1; n <= norms; ++n) { e[p + ".norm" + std::to_string(n) + ".weight"] = {d}; e[p + ".norm" + std::to_string(n) + ".bias"] = {d}; }
    };
    for (int l = 0; l < c->cvae_depth; ++l) {
        const std::string p = "prior_net.encoder.layers." + std::to_string(l);
        attn(p + ".self_attn"); ffn(p, 2);
        const std::string q = "decoder.decoder.layers." + std::to_string(l);
        attn(q + ".self_attn"); attn(q + ".multihead_attn"); ffn(q, 3);
    }
}

int mocha_cvae_load_weight(mocha_ctx* c, const char* name, const float* host, const int64_t* shape, int ndim) {
    if (!c || !name || !host || !shape) return fail(c, MOCHA_ERR_ARG, "null argument");
    cvae_expectations(c);
    const std::string n(name);
    // training-only posterior encoder and the regenerated positional-encoding buffers are accepted and ignored
    if (n.rfind("encoder.", 0) == 0) return 0;
    if (n.size() > 15 && n.compare(n.size() - 15, 15, ".pos_encoder.pe") == 0) {
        // registered buffer (1, max_len, 256): keep its first 192 rows so the table is bit-identical to the checkpoint's
        if (ndim == 3 && shape[0] == 1 && shape[1] >= 192 && shape[2] == 256) {
            HostTensor t; t.data.assign(host, host + 192 * 256); t.shape = {192, 256};
            c->cvae_host["pe"] = std::move(t);
            c->cvae_ready = false;
        }
        return 0;
    }
    auto it = c->cvae_expect.find(n);
    if (it == c->cvae_expect.end()) return fail(c, MOCHA_ERR_WEIGHT, "unknown CVAE weight name '%s'", name);
    if ((int)it->second.size() != ndim) return fail(c, MOCHA_ERR_WEIGHT, "%s: rank %d, expected %zu", name, ndim, it->second.size());
    size_t cnt = 1;
    for (int i = 0; i < ndim; ++i) {
        if (shape[i] != it->second[i]) return fail(c, MOCHA_ERR_WEIGHT, "%s: dim %d is %lld, expected %lld", name, i, (long long)shape[i], (long long)it->second[i]);
        cnt *= (size_t)shape[i];
    }
    HostTensor t; t.data.assign(host, host + cnt); t.shape.assign(shape, shape + ndim);
    c->cvae_host[n] = std::move(t);
    c->cvae_ready = false;
    return 0;
}

int mocha_cvae_finalize(mocha_ctx* c) {
    if (!c) return MOCHA_ERR_ARG;
    cvae_expectations(c);
    for (auto& kv : c->cvae_expect)
        if (!c->cvae_host.count(kv.first)) return fail(c, MOCHA_ERR_STATE, "missing CVAE weight '%s'", kv.first.c_str());
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipDeviceSynchronize());
    x3_drop_images(c);
    h2_drop_images(c);
    for (auto& kv : c->cvae_host) {
        int rc = upload_to(c, c->cw, c->cwsize, kv.first, kv.second.data); if (rc) return rc;
    }
    if (c->cw.count("pe")) { c->cvae_ready = true; return 0; }     // table came with the state_dict (pos_encoder.pe)
    // sin/cos positional encoding rows 0..191 (model_CVAE.py:168-178), float32 arithmetic like torch
    std::vector<float> pe(192 * 256);
    for (int i = 0; i < 128; ++i) {
        const float div = expf((float)(2 * i) * (float)(-std::log(10000.0) / 256.0));
        for (int t = 0; t < 192; ++t) {
            const float a = (float)t * div;
            pe[t * 256 + 2 * i] = sinf(a);
            pe[t * 256 + 2 * i + 1] = cosf(a);
        }
    }
    int rc = upload_to(c, c->cw, c->cwsize, "pe", pe); if (rc) return rc;
    c->cvae_ready = true;
    return 0;
}

static int cvae_ws(mocha_ctx* c, int B) {
    if (c->cvae_B >= B) return 0;
    if (c->cvae_B > 0) HIPCHK(c, hipDeviceSynchronize());
    for (auto& kv : c->cws) dev_free(c, kv.second.p);
    c->cws.clear();
    c->generation++;
    const std::pair<const char*, size_t> plan[] = {{"t0", 182 * 256}, {"t1", 182 * 256}, {"t2", 182 * 256}, {"qkv", 182 * 768},
                                                   {"hff", 182 * 512}, {"mem", 181 * 256}, {"kv", 181 * 512}};
    for (auto& pl : plan) { DevBuf b; b.n = pl.second * (size_t)B; int rc = dev_alloc(c, &b.p, b.n); if (rc) return rc; c->cws[pl.first] = b; }
    c->cvae_B = B;
    return 0;
}

// post-norm transformer sub-blocks shared by PriorNet and Decoder (nn.TransformerEncoderLayer / DecoderLayer defaults)
static int cvae_attn_block(mocha_ctx* c, hipStream_t s, const std::string& p, const std::string& norm, float* x /*in/out*/, int nq,
                           const float* kvsrc /*nullptr: self*/, int nk, int B) {
    float* qkv = c->cws.at("qkv").p; float* t1 = c->cws.at("t1").p; float* t2 = c->cws.at("t2").p; float* kvb = c->cws.at("kv").p;
    const float* Win = c->cw.at(p + ".in_proj_weight"); const float* bin = c->cw.at(p + ".in_proj_bias");
    const int H = c->cvae_heads, DH = 256 / H;
    AttnParams a{};
    if (!kvsrc) {
        GemmParams g = plain(x, 256, Win, qkv, 768, B * nq, 768, 256); g.bias = bin;
        GEMM(c, s, "cvae.in_proj", g);
        a = AttnParams{qkv, qkv + 256, qkv + 512, t1, 768, 768, 768, 256, B, H, DH, nq, nq, (float)std::pow((double)DH, -0.5)};
    } else {
        GemmParams gq = plain(x, 256, Win, qkv, 256, B * nq, 256, 256); gq.bias = bin;
        GEMM(c, s, "cvae.q_proj", gq);
        GemmParams gk = plain(kvsrc, 256, Win + 256 * 256, kvb, 512, B * nk, 512, 256); gk.bias = bin + 256;
        GEMM(c, s, "cvae.kv_proj", gk);
        a = AttnParams{qkv, kvb, kvb + 256, t1, 256, 512, 512, 256, B, H, DH, nq, nk, (float)std::pow((double)DH, -0.5)};
    }
    LAUNCH(c, s, "mocha_attention_f32<64>", "cvae.attn", 4.0 * B * H * (double)a.nq * a.nk * DH, 4.0 * B * (a.nq + 2.0 * a.nk) * 256,
           launch_attention(a, s));
    GemmParams o = plain(t1, 256, c->cw.at(p + ".out_proj.weight"), t2, 256, B * nq, 256, 256);
    o.bias = c->cw.at(p + ".out_proj.bias"); o.residual = x; o.ldr = 256;
    GEMM(c, s, "cvae.out_proj", o);
    LAUNCH(c, s, "mocha_layernorm256", "cvae.norm", 0.0, B * nq * 256.0 * 8,
           launch_layernorm256(t2, c->cw.at(norm + ".weight"), c->cw.at(norm + ".bias"), x, B * nq, s));
    return 0;
}

static int cvae_ff_block(mocha_ctx* c, hipStream_t s, const std::string& p, const std::string& norm, float* x, float* out, int n, int B) {
    float* hff = c->cws.at("hff").p; float* t2 = c->cws.at("t2").p;
    GemmParams f1 = plain(x, 256, c->cw.at(p + ".linear1.weight"), hff, 512, B * n, 512, 256);
    f1.bias = c->cw.at(p + ".linear1.bias"); f1.act = 3;
    GEMM(c, s, "cvae.ff1", f1);
    GemmParams f2 = plain(hff, 512, c->cw.at(p + ".linear2.weight"), t2, 256, B * n, 256, 512);
    f2.bias = c->cw.at(p + ".linear2.bias"); f2.residual = x; f2.ldr = 256;
    GEMM(c, s, "cvae.ff2", f2);
    LAUNCH(c, s, "mocha_layernorm256", "cvae.norm", 0.0, B * n * 256.0 * 8,
           launch_layernorm256(t2, c->cw.at(norm + ".weight"), c->cw.at(norm + ".bias"), out, B * n, s));
    return 0;
}

int mocha_cvae_sample(mocha_ctx* c, const float* cond, int B, float* out, float* mu, float* logvar, const float* eps, void* stream) {
    if (c && B == 0) return 0;
    if (!c || !cond || !out || B < 0) return fail(c, MOCHA_ERR_ARG, "bad CVAE arguments");
    if (!c->cvae_ready) return fail(c, MOCHA_ERR_STATE, "CVAE weights not finalised: call mocha_cvae_finalize first");
    if (B == 0) return 0;
    HIPCHK(c, hipSetDevice(c->device));
    int rc = cvae_ws(c, B); if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    const int nc = c->cvae_nc, nq = c->cvae_nq, ntok = nc + 2;
    float* t0 = c->cws.at("t0").p; float* mem = c->cws.at("mem").p;
    // PriorNet.encode (model_CVAE.py:69-79)
    LAUNCH(c, s, "mocha_cvae_prior_tokens", "cvae.tokens", 0.0, B * ntok * 256.0 * 8,
           launch_cvae_prior_tokens(cond, c->cw.at("prior_net.mu_token"), c->cw.at("prior_net.logvar_token"), c->cw.at("pe"), t0, B, nc, s));
    for (int l = 0; l < c->cvae_depth; ++l) {
        const std::string p = "prior_net.encoder.layers." + std::to_string(l);
        if ((rc = cvae_attn_block(c, s, p + ".self_attn", p + ".norm1", t0, ntok, nullptr, ntok, B))) return rc;
        if ((rc = cvae_ff_block(c, s, p, p + ".norm2", t0, t0, ntok, B))) return rc;
    }
    // reparameterize + Decoder inputs (model_CVAE.py:81-87, 158-163); the decoder state reuses t1's sibling t0 after this
    float* xq = c->cws.at("t1").p;     // free until the first attention writes it: use hff as query buffer instead
    xq = c->cws.at("hff").p;           // (B, nq, 256) fits in (B, 182, 512)
    LAUNCH(c, s, "mocha_cvae_latent", "cvae.latent", 0.0, B * (nc + nq + 2) * 256.0 * 4,
           launch_cvae_latent(t0, ntok, eps, cond, nc, c->cw.at("pe"), nq, mem, xq, mu, logvar, B, s));
    HIPCHK(c, hipMemcpyAsync(t0, xq, (size_t)B * nq * 256 * sizeof(float), hipMemcpyDeviceToDevice, s));
    for (int l = 0; l < c->cvae_depth; ++l) {
        const std::string p = "decoder.decoder.layers." + std::to_string(l);
        if ((rc = cvae_attn_block(c, s, p + ".self_attn", p + ".norm1", t0, nq, nullptr, nq, B))) return rc;
        if ((rc = cvae_attn_block(c, s, p + ".multihead_attn", p + ".norm2", t0, nq, mem, nc + 1, B))) return rc;
        const bool last = l == c->cvae_depth - 1;
        if ((rc = cvae_ff_block(c, s, p, p + ".norm3", t0, last ? out : t0, nq, B))) return rc;
    }
    return 0;
}

int mocha_cvae_condition(mocha_ctx* c, const float* src_cnt, const float* src_mean, const float* src_std, const float* prev_cha,
                         const float* cha_mean, const float* cha_std, int B, float* cond, void* stream) {
    if (c && B == 0) return 0;
    if (!c || !src_cnt || !src_mean || !src_std || !prev_cha || !cha_mean || !cha_std || !cond || B < 0) return fail(c, MOCHA_ERR_ARG, "bad arguments");
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t s = (hipStream_t)stream;
    LAUNCH(c, s, "mocha_cvae_condition", "cvae.condition", 0.0, B * 180.0 * 256 * 8,
           launch_cvae_condition(src_cnt, src_mean, src_std, prev_cha, cha_mean, cha_std, cond, B, 90, s));
    return 0;
}

int mocha_scale_shift(mocha_ctx* c, const float* x, const float* mean, const float* std_, int B, float* out, void* stream) {
    if (c && B == 0) return 0;
    if (!c || !x || !mean || !std_ || !out || B < 0) return fail(c, MOCHA_ERR_ARG, "bad arguments");
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t s = (hipStream_t)stream;
    LAUNCH(c, s, "mocha_scale_shift", "cvae.denorm", 0.0, B * 90.0 * 256 * 8, launch_scale_shift(x, mean, std_, out, B, 90, s));
    return 0;
}

int mocha_featurize(mocha_ctx* c, const float* Yrot, const float* Ypos, const float* Yvel, const float* Yang, int B, float* X_raw,
                    void* stream) {
    if (c && B == 0) return 0;
    if (!c || !Yrot || !Ypos || !Yvel || !Yang || !X_raw || B < 0) return fail(c, MOCHA_ERR_ARG, "bad featurize arguments");
    HIPCHK(c, hipSetDevice(c->device));
    const int J = c->cfg.V + 1;
    if (!c->bone_parents) {                       // parents = [-1] + (joint parents + 1), test_fullframework.py:101-102
        std::vector<int> par(J);
        par[0] = -1;
        for (int i = 0; i < c->cfg.V; ++i) par[i + 1] = c->sk.parents[i] + 1;
        void* dp = nullptr;
        HIPCHK(c, hipMalloc(&dp, sizeof(int) * J));
        HIPCHK(c, hipMemcpy(dp, par.data(), sizeof(int) * J, hipMemcpyHostToDevice));
        c->bone_parents = (int*)dp;
    }
    hipStream_t s = (hipStream_t)stream;
    LAUNCH(c, s, "mocha_featurize", "featurize", B * 60.0 * J * 150, B * 60.0 * J * (13 + 15) * 4,
           launch_featurize(Yrot, Ypos, Yvel, Yang, c->bone_parents, X_raw, B, c->cfg.T, J, s));
    return 0;
}

int mocha_pose_heads(mocha_ctx* c, const float* Y, int B, float* heads, float* speed, void* stream) {
    if (c && B == 0) return 0;
    if (!c || !Y || !heads || !speed || B < 0) return fail(c, MOCHA_ERR_ARG, "bad pose_heads arguments");
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t s = (hipStream_t)stream;
    LAUNCH(c, s, "mocha_pose_heads", "post.heads", 0.0, B * (60.0 * 12 + c->cfg.V * 28.0 * 4),
           launch_pose_heads(Y, heads, speed, B, c->cfg.T, c->cfg.V, s));
    return 0;
}

void mocha_post_cfg_default(mocha_post_cfg* cfg) {
    if (!cfg) return;
    *cfg = mocha_post_cfg{};
    cfg->dt = 1.0 / 60.0;                   // test_fullframework.py:105
    cfg->ik_max_length_buffer = 0.015;      // :109-114
    cfg->ik_foot_height = 0.02;
    cfg->ik_unlock_radius = 0.2;
    cfg->ik_blending_halflife = 0.1;
    cfg->ik_enabled = 1;
    cfg->blend_enabled = 1;
    cfg->n_contact = 2;                     // :104
    cfg->contact_bones[0] = 5;
    cfg->contact_bones[1] = 24;
}

int mocha_postprocess(mocha_ctx* c, const mocha_post_cfg* cfg, const float* heads, const float* speed, const float* src_rvel,
                      const float* src_rang, const float* src_speed, const unsigned char* contact, int n_clips, int n_frames,
                      double* pos, double* rot, double* ik_rot, double* bvh_pos, double* bvh_euler, void* stream) {
    if (c && (n_clips == 0 || n_frames == 0)) return 0;
    if (!c || !heads || !speed || !src_rvel || !src_rang || !src_speed || !contact || !pos || !rot || !ik_rot || n_clips < 0 ||
        n_frames < 0 || (!bvh_pos) != (!bvh_euler))
        return fail(c, MOCHA_ERR_ARG, "bad postprocess arguments");
    mocha_post_cfg d;
    if (!cfg) { mocha_post_cfg_default(&d); cfg = &d; }
    const int J = c->cfg.V + 1;
    if (J > MOCHA_MAX_BONES) return fail(c, MOCHA_ERR_ARG, "postprocess: too many bones");
    if (cfg->n_contact < 0 || cfg->n_contact > MOCHA_MAX_CONTACT) return fail(c, MOCHA_ERR_ARG, "postprocess: n_contact must be 0..4");
    if (!(cfg->dt > 0.0)) return fail(c, MOCHA_ERR_ARG, "postprocess: dt must be positive");
    PostParams p{};
    p.heads = heads; p.speed = speed; p.src_rvel = src_rvel; p.src_rang = src_rang; p.src_speed = src_speed; p.contact = contact;
    p.pos = pos; p.rot = rot; p.ik_rot = ik_rot; p.bvh_pos = bvh_pos; p.bvh_euler = bvh_euler;
    p.n_clips = n_clips; p.n_frames = n_frames; p.V = c->cfg.V; p.n_contact = cfg->n_contact; p.ik_enabled = cfg->ik_enabled;
    p.blend_enabled = cfg->blend_enabled;
    p.parents[0] = -1;                       // parents = [-1] + (joint parents + 1), test_fullframework.py:101-102
    for (int i = 0; i < c->cfg.V; ++i) p.parents[i + 1] = c->sk.parents[i] + 1;
    for (int i = 0; i < cfg->n_contact; ++i) {
        const int toe = cfg->contact_bones[i];
        if (toe < 1 || toe >= J) return fail(c, MOCHA_ERR_ARG, "postprocess: contact bone out of range");
        int depth = 0;
        for (int b = toe; b != -1; b = p.parents[b]) ++depth;
        if (depth < 5 || depth > MOCHA_MAX_CHAIN) return fail(c, MOCHA_ERR_ARG, "postprocess: contact bone needs 4..7 ancestors");
        p.contact_bones[i] = toe;
    }
    p.dt = cfg->dt; p.max_length_buffer = cfg->ik_max_length_buffer; p.foot_height = cfg->ik_foot_height;
    p.unlock_radius = cfg->ik_unlock_radius; p.halflife = cfg->ik_blending_halflife;
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t s = (hipStream_t)stream;
    LAUNCH(c, s, "mocha_post_clip", "post.clip", 0.0, (double)n_clips * n_frames * (c->cfg.V * 13.0 * 4 + J * 11.0 * 8),
           launch_post_clip(p, s));
    return 0;
}

int mocha_column_stats(mocha_ctx* c, const float* x, int64_t N, float* mean, float* std_, void* stream) {
    if (!c || !x || !mean || !std_ || N < 1) return fail(c, MOCHA_ERR_ARG, "bad column_stats arguments");
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t s = (hipStream_t)stream;
    LAUNCH(c, s, "mocha_column_stats", "bank.stats", 0.0, 8.0 * N * 23040, launch_column_stats(x, N, 90 * 256, mean, std_, s));
    return 0;
}

int mocha_set_option(mocha_ctx* c, const char* name, int value) {
    if (!c || !name) return MOCHA_ERR_ARG;
    const std::string n(name);
    if (n == "dual_stream") {
        HIPCHK(c, hipSetDevice(c->device));
        HIPCHK(c, hipDeviceSynchronize());
        c->dual_stream = value != 0;
        c->chunk = 0;                              // workspaces are re-planned (one or two sets) on the next call
        return 0;
    }
    if (n == "lanes") {
        if (value < 1 || value > mocha_ctx::MAX_SETS) return fail(c, MOCHA_ERR_ARG, "lanes must be 1..%d", mocha_ctx::MAX_SETS);
        HIPCHK(c, hipSetDevice(c->device));
        HIPCHK(c, hipDeviceSynchronize());
        c->lanes = value;
        c->chunk = 0;                              // workspace sets are re-planned on the next call (generation moves)
        return 0;
    }
    if (n == "dual_min") {
        HIPCHK(c, hipSetDevice(c->device));
        HIPCHK(c, hipDeviceSynchronize());
        c->dual_min = value < 2 ? 2 : value;
        c->chunk = 0;                              // whether set 1 is a full-size set depends on it: re-planned on the next call
        return 0;
    }
    // Path-selecting options change which kernels a step launches: a captured step graph (mocha_step_graph_lane, OursSession)
    // replays the old path until it is re-captured, so every one of them moves the generation.
    if (n == "fold_decoder") { c->fold_decoder = value != 0; c->generation++; return 0; }
    if (n == "upsample_split_min") { c->upsample_split_min = value; c->generation++; return 0; }
    if (n == "fold_upsample") { c->fold_upsample = value != 0; c->generation++; return 0; }
    if (n == "fold_joint") { c->fold_joint = value != 0; c->generation++; return 0; }
    if (n == "match_fold") { c->match_fold = value != 0; c->generation++; return 0; }
    if (n == "scan8") { c->scan8 = value != 0; if (!value) c->bank8_valid = false; c->generation++; return 0; }     // the image is built at the next mocha_bank_set
    if (n == "scan16") { c->scan16 = value != 0; c->generation++; return 0; }             // bank side takes effect at the next mocha_bank_set
    if (n == "attention_split_max") { c->attn_split_max = value < 0 ? 0 : value; c->generation++; return 0; }      // this context only
    if (n == "gemm_persistent_max_n") { c->gemm_persistent_max_n = value < 128 ? 128 : value; c->generation++; return 0; }
    if (n == "gemm_x3r_min_n") { c->gemm_x3r_min_n = value < 0 ? 0 : value; c->generation++; return 0; }
    if (n == "gemm_tile64_below") { c->gemm_tile64_below = value < 0 ? 0 : value; c->generation++; return 0; }
    if (n == "gemm_persistent") { c->gemm_persistent = value < 0 ? 0 : (value + 7) / 8 * 8; c->generation++; return 0; }
    if (n == "embed_sums") { c->embed_sums = value != 0; c->generation++; return 0; }
    if (n == "embed_front_max_wgs") { c->embed_max_wgs = value; c->generation++; return 0; }
    if (n == "inorm_split_max") { c->inorm_split_max = value < 0 ? 0 : value; c->generation++; return 0; }
    if (n == "pair_overlap") { c->pair_overlap = value != 0; c->generation++; return 0; }
    if (n == "adain_closed_form") { c->adain_closed = value != 0; c->generation++; return 0; }
    if (n == "style_f64") { c->style_f64 = value != 0; c->bank_dec_valid = false; c->generation++; return 0; }     // cached bank constants: rebuilt at the next mocha_bank_set
    if (n == "bank_dec_cache") { c->bank_dec_cache = value != 0; if (!value) c->bank_dec_valid = false; c->generation++; return 0; }     // takes effect at the next mocha_bank_set
    if (n == "bank_tiled") { c->use_tiled = value != 0; c->generation++; return 0; }
    if (n == "match_planes") { if (value != 1 && value != 2) return fail(c, MOCHA_ERR_ARG, "match_planes must be 1 or 2"); c->match_planes = value; c->generation++; return 0; }
    if (n == "select2") { c->select2 = value != 0; c->generation++; return 0; }
    if (n == "match_pass") { if (value < 0 || value > 2) return fail(c, MOCHA_ERR_ARG, "match_pass must be 0, 1 or 2"); c->match_pass = value; c->generation++; return 0; }
    if (n == "match_nt") { c->match_nt = value != 0; c->generation++; return 0; }
    if (n == "match_pass_variant") { c->match_pass_variant = value & 31; c->generation++; return 0; }      // prefetch depth + non-temporal bit; the fill-only floor kernel (bit 256: no multiplies) is tools/match_pass_probe's, never the library's
    if (n == "match_pass_max_q") { c->match_pass_max_q = value; c->generation++; return 0; }
    if (n == "attention_kv") {
        HIPCHK(c, hipSetDevice(c->device));
        HIPCHK(c, hipDeviceSynchronize());
        c->attn_kv = value != 0;
        c->chunk = 0;                              // the key / value image workspace exists only while the option is on: re-planned on the next call
        return 0;
    }
    if (n == "attention_kv_pairs") { c->attn_kv_pairs = value != 0; c->generation++; return 0; }
    if (n == "gemm_bf16x3") { c->gemm_x3 = value != 0; c->generation++; return 0; }
    if (n == "gemm_f16x2") {                        // encoder / decoder / to_mot GEMMs on two fp16 planes, three passes (gemm_h2.hip); the bank's bound: next mocha_bank_set
        HIPCHK(c, hipSetDevice(c->device));
        if (value) { int rc = amax_alloc(c); if (rc) return rc; }
        c->gemm_h2 = value != 0; c->amax_bank_ok = false;
        for (int i = 0; i < mocha_ctx::MAX_SETS; ++i) c->amax_enc_of[i] = c->amax_dec_of[i] = c->amax_tok_of[i] = nullptr;
        c->generation++; return 0;
    }
    if (n == "attention_bf16x3") { c->attn_x3 = value != 0; c->generation++; return 0; }
    return fail(c, MOCHA_ERR_ARG, "unknown option '%s'", name);
}

int mocha_linear(mocha_ctx* c, const float* x, const float* w, const float* bias, float* y, int64_t M, int N, int K, int engine,
                 void* stream) {
    if (!c || !x || !w || !y) return fail(c, MOCHA_ERR_ARG, "null argument");
    if (M < 0 || N < 1 || K < 32 || K % 32 != 0 || M > (1ll << 30) || engine < 0 || engine > 3)
        return fail(c, MOCHA_ERR_ARG, "mocha_linear: M=%lld N=%d K=%d engine=%d (K %% 32 == 0, engine 0..3)", (long long)M, N, K, engine);
    if (M == 0) return 0;
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t s = (hipStream_t)stream;
    GemmParams p = plain(x, K, w, y, N, (int)M, N, K);
    p.bias = bias;
    if (engine == 3) {                              // two fp16 planes, three passes: the activation bound is measured here (mocha_absmax)
        p.rows_per_win = 1024;
        if (!gemm_h2_supports(p)) return fail(c, MOCHA_ERR_ARG, "mocha_linear: M=%lld N=%d K=%d is outside the f16x2 engine", (long long)M, N, K);
        void* img = nullptr; float* aux = nullptr;       // aux: [N] inverse weight scales, then the activation bounds of the "windows" (blocks of 1 024 rows here)
        HIPCHK(c, hipMalloc(&img, gemm_h2_packed_elems(N, K) * sizeof(unsigned short)));
        const size_t n4 = ((size_t)N + 3) / 4 * 4;
        const int rpw = 1024; const long long nwin = (M + rpw - 1) / rpw;
        hipError_t e = hipMalloc((void**)&aux, (n4 + (size_t)nwin) * sizeof(float));
        if (e == hipSuccess) e = hipMemsetAsync(aux + n4, 0, (size_t)nwin * sizeof(float), s);
        if (e == hipSuccess) e = launch_pack_h2(w, N, K, (unsigned short*)img, aux, s);
        for (long long w0 = 0; w0 < nwin && e == hipSuccess; w0 += 65535) {
            const long long nw = std::min<long long>(65535, nwin - w0);
            // the last block may be short: measured on its own
            const long long full = (w0 + nw == nwin && M % rpw) ? nw - 1 : nw;
            if (full > 0) e = launch_absmax(x + (size_t)w0 * rpw * K, full, (long long)rpw * K, aux + n4 + w0, s);
            if (e == hipSuccess && full < nw) e = launch_absmax(x + (size_t)(w0 + full) * rpw * K, 1, (long long)(M % rpw) * K, aux + n4 + w0 + full, s);
        }
        p.Wh2 = (const unsigned short*)img; p.w_inv = aux; p.a_amax = aux + n4; p.rows_per_win = rpw;
        if (e == hipSuccess) e = launch_gemm_h2(p, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);
        (void)hipFree(img); if (aux) (void)hipFree(aux);
        if (e != hipSuccess) return fail(c, MOCHA_ERR_HIP, "mocha_linear: %s", hipGetErrorString(e));
        return 0;
    }
    const bool x3 = engine == 2 || (engine == 0 && c->gemm_x3 && gemm_x3_supports(p));
    if (!x3) {
        const double flops = 2.0 * M * (double)N * K;
        LAUNCH(c, s, gemm_kernel_name(p), "linear", flops, 4.0 * ((double)M * K + (double)N * K + (double)M * N), launch_gemm(p, s));
        return 0;
    }
    if (!gemm_x3_supports(p)) return fail(c, MOCHA_ERR_ARG, "mocha_linear: M=%lld N=%d K=%d is outside the bf16x3 engine", (long long)M, N, K);
    void* img = nullptr;
    HIPCHK(c, hipMalloc(&img, gemm_x3_packed_elems(N, K) * sizeof(unsigned short)));
    hipError_t e = launch_pack_x3(w, N, K, (unsigned short*)img, s);
    p.Wsplit = (const unsigned short*)img; p.persistent = c->gemm_persistent; p.persistent_max_n = c->gemm_persistent_max_n;
    if (e == hipSuccess) e = (c->gemm_x3r_min_n > 0 && N >= c->gemm_x3r_min_n && gemm_x3r_supports(p)) ? launch_gemm_x3r(p, s) : launch_gemm_x3(p, s);      // option "gemm_x3r_min_n": same bits
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    (void)hipFree(img);
    if (e != hipSuccess) return fail(c, MOCHA_ERR_HIP, "mocha_linear: %s", hipGetErrorString(e));
    return 0;
}

int mocha_scan_byte_state(mocha_ctx* c, int set, int32_t* state, void* stream) {
    if (!c || !state || set < 0 || set >= mocha_ctx::MAX_SETS) return fail(c, MOCHA_ERR_ARG, "mocha_scan_byte_state: bad argument");
    HIPCHK(c, hipSetDevice(c->device));
    state[0] = c->scan8 && c->bank8_valid ? 1 : 0;
    state[1] = -1;
    if (c->scan_keys[set]) {
        unsigned w[2] = {0, 0};
        (void)stream;
        HIPCHK(c, hipDeviceSynchronize());                     // introspection: whatever stream the calls ran on
        HIPCHK(c, hipMemcpy(w, c->scan_keys[set] + match_scan8_mode_word(), sizeof(w), hipMemcpyDeviceToHost));
        state[1] = (int32_t)w[1];
    }
    return 0;
}

int mocha_bank_view(mocha_ctx* c, const float** cnt_nm, const float** encoded, const float** centroid, const float** row_norm2,
                    const void** cnt_bf16, int64_t* N) {
    if (!c) return MOCHA_ERR_ARG;
    if (!c->bank_cnt || c->bank_N <= 0) return fail(c, MOCHA_ERR_STATE, "no bank: call mocha_bank_set first");
    if (cnt_nm) *cnt_nm = c->bank_cnt;
    if (encoded) *encoded = c->bank_enc;
    if (centroid) *centroid = c->bank_center;
    if (row_norm2) *row_norm2 = c->bank_norm;
    if (cnt_bf16) *cnt_bf16 = c->bank_is_bf16 ? c->bank_bf16 : nullptr;
    if (N) *N = c->bank_N;
    return 0;
}

int mocha_bank_export(mocha_ctx* c, float* cnt_nm, float* encoded, void* stream) {
    int rc = ready(c, 0); if (rc) return rc;
    if (!c->bank_cnt || !c->bank_enc || c->bank_N <= 0) return fail(c, MOCHA_ERR_STATE, "no bank: call mocha_bank_set or mocha_bank_broadcast first");
    const size_t bytes = (size_t)c->bank_N * 90 * 256 * sizeof(float);
    hipStream_t s = (hipStream_t)stream;
    if (cnt_nm) HIPCHK(c, hipMemcpyAsync(cnt_nm, c->bank_cnt, bytes, hipMemcpyDeviceToDevice, s));
    if (encoded) HIPCHK(c, hipMemcpyAsync(encoded, c->bank_enc, bytes, hipMemcpyDeviceToDevice, s));
    return 0;
}

int mocha_profile_start(mocha_ctx* c) {
    if (!c) return MOCHA_ERR_ARG;
    for (auto& r : c->prof) { (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1); }
    c->prof.clear();
    c->prof_on = true;
    return 0;
}

int mocha_profile_stop(mocha_ctx* c, char* json, int64_t cap) {
    if (!c || !json || cap < 64) return fail(c, MOCHA_ERR_ARG, "bad profile buffer");
    c->prof_on = false;
    HIPCHK(c, hipSetDevice(c->device));
    struct Agg { long n = 0; double ms = 0, flops = 0, bytes = 0; };
    std::map<std::string, Agg> byk, bys;
    for (auto& r : c->prof) {
        HIPCHK(c, hipEventSynchronize(r.e1));
        float ms = 0.f;
        HIPCHK(c, hipEventElapsedTime(&ms, r.e0, r.e1));
        for (auto* m : {&byk, &bys}) {
            Agg& a = (*m)[m == &byk ? r.kernel : r.site + "|" + r.kernel];
            a.n++; a.ms += ms; a.flops += r.flops; a.bytes += r.bytes;
        }
        (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1);
    }
    c->prof.clear();
    std::string o = "{";
    for (int pass = 0; pass < 2; ++pass) {
        o += pass ? ",\"sites\":{" : "\"kernels\":{";
        bool first = true;
        for (auto& kv : (pass ? bys : byk)) {
            char buf[384];
            snprintf(buf, sizeof buf, "%s\"%s\":{\"launches\":%ld,\"ms\":%.6f,\"flops\":%.6e,\"bytes\":%.6e}", first ? "" : ",",
                     kv.first.c_str(), kv.second.n, kv.second.ms, kv.second.flops, kv.second.bytes);
            o += buf; first = false;
        }
        o += "}";
    }
    o += "}";
    if ((int64_t)o.size() + 1 > cap) return fail(c, MOCHA_ERR_ARG, "profile buffer too small (%zu needed)", o.size() + 1);
    memcpy(json, o.c_str(), o.size() + 1);
    return 0;
}

}  // extern "C"
