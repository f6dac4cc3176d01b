"""CPU ORACLE for the MOCHA Generator hot path — TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this module.  The product path (``mocha_sigasia2023_amd``) never does; it fails
loudly when the HIP library is missing.

This is a plain PyTorch-CPU fp32 functional restatement (from a raw ``state_dict``-shaped
mapping, no ``nn.Module``) of the arithmetic of the reference path, op for op and in the
reference's own op order (NCHW convs, einsums, reflect padding), so that it is both the
parity checker and a fair stand-in for "the reference CPU PyTorch path" when timed
(SURVEY.md §8c/§8d).  Every function cites the reference file:line it follows
(paths relative to the reference repository root).

PARITY PINNING: the reference ships no tests or golden vectors for this path
(SURVEY.md §4).  The oracle is pinned against outputs of the reference itself, generated
in the build container by ``tests/golden/make_golden.py`` (which imports the reference
from /root/reference, loads the same synthetic weights, and stores inputs + per-stage
outputs as fixtures under ``tests/golden/``); ``tests/test_oracle_golden.py`` checks this
module against those fixtures.  The nearest-neighbour search follows scikit-learn
``BallTree(k=1)`` semantics (un-vendored, un-pinned dependency, environment.yml:15;
fixtures made with scikit-learn 1.7.2): exact Euclidean 1-NN in float64.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F

LRELU_SLOPE = 0.2   # nn.LeakyReLU(0.2): net/blocks.py:96, model.py:77, net/transformer.py:105


def _t(sd, name):
    v = sd[name]
    if isinstance(v, np.ndarray):
        v = torch.from_numpy(v)
    return v.detach().to(torch.float32)


def to_torch_state(sd):
    return {k: _t(sd, k) for k in sd}


# --------------------------------------------------------------------------- blocks
def mean_variance_norm(x, eps=1e-5):
    """net/transformer.py:13-20 — x (B, C, S): per (b, c) over S, unbiased std, eps outside sqrt."""
    size = x.size()
    x = x.reshape(size[0], size[1], -1)
    mean = x.mean(-1, keepdim=True)
    std = x.std(-1, keepdim=True)
    return ((x - mean) / (std + eps)).reshape(size)


def cnt_feature(encoded):
    """test_fullframework.py:193-194 / model.py:96-98 — 'cnt' of encoded (B, N, C)."""
    return mean_variance_norm(encoded.permute(0, 2, 1)).permute(0, 2, 1).contiguous()


def spatial_conv(x, A, w, b):
    """SpatialConv.forward, net/blocks.py:57-66 — 1x1 conv to K*C then einsum with A (K,V,V)."""
    K = A.shape[0]
    x = F.conv2d(x, w, b)
    n, kc, t, v = x.shape
    x = x.view(n, K, kc // K, t, v)
    return torch.einsum('nkctv,kvw->nctw', x, A).contiguous()


def stgcn_block(x, A, gw, gb, tw, tb):
    """STGCN_Block.forward, net/blocks.py:125-134 (norm='none', activation='lrelu', t_padding=True):
    LeakyReLU -> gcn -> temporal conv (k_t, 1) with reflect padding (net/blocks.py:112-118)."""
    x = F.leaky_relu(x, LRELU_SLOPE)
    x = spatial_conv(x, A, gw, gb)
    pad = (tw.shape[2] - 1) // 2
    x = F.pad(x, (0, 0, pad, pad), mode='reflect')
    return F.conv2d(x, tw, tb)


def mot_embedding(sd, X, stages=None):
    """Generator.mot_embedding, model.py:42-50.  X (B, T, V, C_in) -> tokens (B, 90, 256)."""
    x = X.permute(0, 3, 1, 2).contiguous()                                    # model.py:43
    x = F.conv2d(x, sd['mot_embedding.1.weight'], sd['mot_embedding.1.bias'])  # model.py:44
    if stages is not None:
        stages['emb_conv1'] = x
    x = stgcn_block(x, sd['mot_embedding.2.A_j'],                              # model.py:45,109-134
                    sd['mot_embedding.2.blk.gcn.conv.weight'], sd['mot_embedding.2.blk.gcn.conv.bias'],
                    sd['mot_embedding.2.blk.tcn.weight'], sd['mot_embedding.2.blk.tcn.bias'])
    if stages is not None:
        stages['emb_joint'] = x
    x = torch.einsum('nctv,vw->nctw', x, sd['mot_embedding.3.weight'])        # net/graph.py:463-465
    x = F.avg_pool2d(x, kernel_size=(4, 1))                                   # model.py:47
    if stages is not None:
        stages['emb_pooled'] = x
    x = stgcn_block(x, sd['mot_embedding.5.A_b'],                              # model.py:48,137-162
                    sd['mot_embedding.5.blk.gcn.conv.weight'], sd['mot_embedding.5.blk.gcn.conv.bias'],
                    sd['mot_embedding.5.blk.tcn.weight'], sd['mot_embedding.5.blk.tcn.bias'])
    b, c, t, v = x.shape
    return x.permute(0, 2, 3, 1).reshape(b, t * v, c).contiguous()            # model.py:49


def attention(sd, p, src, tar, heads, adain):
    """Attention.forward, net/transformer.py:63-76 (+ mapping_function :49-56)."""
    if adain:
        q_in = mean_variance_norm(src.permute(0, 2, 1)).permute(0, 2, 1)
        k_in = mean_variance_norm(tar.permute(0, 2, 1)).permute(0, 2, 1)
    else:
        q_in, k_in = src, tar
    q = F.linear(q_in, sd[f'{p}.to_q.1.weight'])
    k = F.linear(k_in, sd[f'{p}.to_k.1.weight'])
    v = F.linear(tar, sd[f'{p}.to_v.weight'])
    b, n, inner = q.shape
    dh = inner // heads
    q, k, v = (t.view(b, -1, heads, dh).permute(0, 2, 1, 3) for t in (q, k, v))
    dots = torch.matmul(q, k.transpose(-1, -2)) * (dh ** -0.5)
    attn = dots.softmax(dim=-1)
    out = torch.matmul(attn, v).permute(0, 2, 1, 3).reshape(b, n, inner)
    return F.linear(out, sd[f'{p}.to_out.0.weight'], sd[f'{p}.to_out.0.bias'])


def feed_forward(sd, p, x):
    """FeedForward, net/transformer.py:23-34 — Linear, exact-erf GELU, Linear (dropout off in eval)."""
    h = F.gelu(F.linear(x, sd[f'{p}.net.0.weight'], sd[f'{p}.net.0.bias']))
    return F.linear(h, sd[f'{p}.net.3.weight'], sd[f'{p}.net.3.bias'])


def adain(sd, p, x, sty):
    """AdaIN.forward, net/transformer.py:98-113."""
    s = sty.permute(0, 2, 1).mean(-1)                                  # AdaptiveAvgPool1d(1) + Rearrange
    s = F.linear(s, sd[f'{p}.style.2.weight'], sd[f'{p}.style.2.bias'])
    s = F.leaky_relu(s, LRELU_SLOPE)
    s = F.linear(s, sd[f'{p}.style.4.weight'], sd[f'{p}.style.4.bias']).unsqueeze(2)
    gamma, beta = s.chunk(2, 1)
    out = mean_variance_norm(x.permute(0, 2, 1))
    out = (1 + gamma) * out + beta
    return out.permute(0, 2, 1)


def transformer(sd, name, x, sty=None, depth=2, heads=4, stages=None):
    """Transformer.forward, net/transformer.py:90-95.  adain iff sty is given (decoder)."""
    use_adain = sty is not None
    for l in range(depth):
        p = f'{name}.layers.{l}'
        if use_adain:
            x = adain(sd, f'{p}.0', x, sty)
        x = attention(sd, f'{p}.1', x, sty if use_adain else x, heads, use_adain) + x
        x = feed_forward(sd, f'{p}.2', x) + x
        if stages is not None:
            stages[f'{name}_l{l}'] = x
    return x


# configs/config.yaml:19,25 (encoder_heads / decoder_heads): the one model dimension the weights' shapes do not determine (depth is
# counted from the state_dict, dim_head = inner / heads).  Tests of non-default configurations set it through heads_config().
HEADS = {'encoder': 4, 'decoder': 4}


class heads_config:
    """with heads_config(enc, dec): ... - the oracle's encoder / decoder head counts inside the block (model.py:53-68)."""

    def __init__(self, enc=4, dec=4):
        self.new = {'encoder': enc, 'decoder': dec}

    def __enter__(self):
        self.old = dict(HEADS); HEADS.update(self.new)

    def __exit__(self, *a):
        HEADS.update(self.old)


def encoder(sd, tokens, stages=None):
    """model.py:53-59 — Transformer(dim 256, depth 2, heads 4, dim_head 128, mlp 512, adain=False)."""
    return transformer(sd, 'encoder', tokens, None, depth=_depth(sd, 'encoder'), heads=HEADS['encoder'], stages=stages)


def decoder(sd, src_enc, cha_enc, stages=None):
    """model.py:62-68 — Transformer(dim 256, depth 2, heads 4, dim_head 256, mlp 512, adain=True)."""
    return transformer(sd, 'decoder', src_enc, cha_enc, depth=_depth(sd, 'decoder'), heads=HEADS['decoder'], stages=stages)


def _depth(sd, name):
    d = 0
    while f'{name}.layers.{d}.1.to_v.weight' in sd:
        d += 1
    return d


def to_mot(sd, tokens, stages=None):
    """Generator.to_mot, model.py:71-80.  tokens (B, 90, 256) -> (B, T, V, C_in)."""
    b, n, c = tokens.shape
    x = tokens.view(b, n // 6, 6, c).permute(0, 3, 1, 2).contiguous()         # model.py:72
    x = stgcn_block(x, sd['to_mot.1.A_b'],                                     # model.py:73
                    sd['to_mot.1.blk.gcn.conv.weight'], sd['to_mot.1.blk.gcn.conv.bias'],
                    sd['to_mot.1.blk.tcn.weight'], sd['to_mot.1.blk.tcn.bias'])
    if stages is not None:
        stages['mot_body'] = x
    x = F.interpolate(x, scale_factor=(4, 1), mode='nearest')                  # model.py:74,165-174
    x = torch.einsum('nctv,vw->nctw', x, sd['to_mot.3.weight'])                # net/graph.py:606-608
    x = stgcn_block(x, sd['to_mot.4.A_j'],                                     # model.py:76
                    sd['to_mot.4.blk.gcn.conv.weight'], sd['to_mot.4.blk.gcn.conv.bias'],
                    sd['to_mot.4.blk.tcn.weight'], sd['to_mot.4.blk.tcn.bias'])
    if stages is not None:
        stages['mot_joint'] = x
    x = F.leaky_relu(x, LRELU_SLOPE)                                           # model.py:77
    x = F.conv2d(x, sd['to_mot.6.weight'], sd['to_mot.6.bias'])                # model.py:78
    return x.permute(0, 2, 3, 1).contiguous()                                  # model.py:79


def generator_forward(sd, src_X, cha_X, extract_feature=False):
    """Generator.forward, model.py:82-106."""
    src_tokens = mot_embedding(sd, src_X)
    cha_tokens = mot_embedding(sd, cha_X)
    src_tokens = src_tokens + sd['pos_emb'][:, :src_tokens.shape[1]]
    cha_tokens = cha_tokens + sd['pos_emb'][:, :cha_tokens.shape[1]]
    src_encoded = encoder(sd, src_tokens)
    cha_encoded = encoder(sd, cha_tokens)
    if extract_feature:
        return src_encoded, cha_encoded, cnt_feature(src_encoded), cnt_feature(cha_encoded)
    return to_mot(sd, decoder(sd, src_encoded, cha_encoded))


def encode(sd, X):
    """The demo's encode sequence, test_fullframework.py:190-193: tokens, +pos_emb, encoder, cnt."""
    tokens = mot_embedding(sd, X)
    tokens = tokens + sd['pos_emb'][:, :tokens.shape[1]]
    encoded = encoder(sd, tokens)
    return encoded, cnt_feature(encoded)


# --------------------------------------------------------------------------- context matching
def temporal_weight(num_temp=15, nbody=6, dim=256):
    """train_CVAE.py:64-66 — linspace(1, 3, 15) per time-patch, broadcast over (body part, channel)."""
    w = np.linspace(1.0, 3.0, num_temp, dtype=np.float32)
    return np.repeat(w, nbody)[:, None].repeat(dim, 1).astype(np.float32)     # (90, 256)


def znorm(cnt, cnt_mean, cnt_std):
    """test_fullframework.py:293,297,442 — (cnt - mean) / std with std already divided by temp_weight (:89)."""
    return (cnt - cnt_mean[None]) / cnt_std[None]


def match_bruteforce(query_nm, bank_nm):
    """Exact Euclidean 1-NN in float64 — the semantics of BallTree(bank).query(q, k=1)
    (test_fullframework.py:294-296,443; scikit-learn up-casts to float64).
    query_nm (Q, ...), bank_nm (N, ...) -> (idx int64 (Q,), dist float64 (Q,))."""
    q = np.asarray(query_nm, dtype=np.float64).reshape(len(query_nm), -1)
    k = np.asarray(bank_nm, dtype=np.float64).reshape(len(bank_nm), -1)
    idx = np.empty(len(q), dtype=np.int64)
    dist = np.empty(len(q), dtype=np.float64)
    kk = (k * k).sum(1)
    for s in range(0, len(q), 64):
        qq = q[s:s + 64]
        d2 = (qq * qq).sum(1)[:, None] - 2.0 * qq @ k.T + kk[None]
        j = d2.argmin(1)
        # re-evaluate the winners in the direct form (no cancellation)
        dd = ((qq - k[j]) ** 2).sum(1)
        idx[s:s + 64] = j
        dist[s:s + 64] = np.sqrt(dd)
    return idx, dist


def characterize(sd, src_X, cha_X, cnt_mean, cnt_std, batch=32):
    """The NN ('cm_') branch of the demo end to end, test_fullframework.py:188-194,271-277,
    288-302,438-443,465-467, batched: encode both clips (batch 32 like
    collect_CVAE_feature_action.py:167-180), z-score, 1-NN match, gather, decode, to_mot.
    Returns (Y (B_src, T, V, C), idx)."""
    def enc_all(X):
        e, c = [], []
        for s in range(0, len(X), batch):
            a, b = encode(sd, X[s:s + batch])
            e.append(a); c.append(b)
        return torch.cat(e), torch.cat(c)
    src_enc, src_cnt = enc_all(src_X)
    cha_enc, cha_cnt = enc_all(cha_X)
    q = znorm(src_cnt.numpy(), cnt_mean, cnt_std)
    k = znorm(cha_cnt.numpy(), cnt_mean, cnt_std)
    idx, _ = match_bruteforce(q, k)
    sel = cha_enc[torch.from_numpy(idx)]
    out = []
    for s in range(0, len(src_X), batch):
        out.append(to_mot(sd, decoder(sd, src_enc[s:s + batch], sel[s:s + batch])))
    return torch.cat(out), idx
