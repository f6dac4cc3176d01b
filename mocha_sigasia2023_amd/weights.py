"""Weight schema of the reference ``Generator`` (the ``gen_ema`` state_dict the demo loads,
trainer.py:224-247, test_fullframework.py:48-49) and a deterministic synthetic generator.

The pretrained checkpoint is not in the reference repo (download.sh:3-14), so parity and
bench runs use synthetic weights with the exact key names and shapes of the reference
module tree.  The generator is pure numpy and seed-driven so that the very
same tensors can be rebuilt on the GPU box, loaded into the reference model in the build
container (tests/golden/make_golden.py) and into the HIP context.
"""
from __future__ import annotations

from collections import OrderedDict

import numpy as np

from .skeleton import skeleton_constants

# model dims: configs/config.yaml:13-31
DEFAULT_CFG = dict(
    mot_in_dim=15, nframes=60, njoints=24, nbody=6, temporal_patch_size=4,
    encoder_dim=256, encoder_depth=2, encoder_heads=4, encoder_dim_head=128, encoder_mlp_dim=512,
    decoder_dim=256, decoder_depth=2, decoder_heads=4, decoder_dim_head=256, decoder_mlp_dim=512,
)


def param_shapes(cfg=None, V: int = 24) -> "OrderedDict[str, tuple]":
    """name -> shape for every learnable parameter, in the reference's registration order."""
    c = dict(DEFAULT_CFG, **(cfg or {}))
    d = c["encoder_dim"]
    d4 = d // c["temporal_patch_size"]
    cin = c["mot_in_dim"]
    ntok = c["nbody"] * (c["nframes"] // c["temporal_patch_size"])
    s = OrderedDict()
    s["pos_emb"] = (1, ntok, d)                                           # model.py:40
    s["mot_embedding.1.weight"] = (d4, cin, 1, 1)                         # model.py:44
    s["mot_embedding.1.bias"] = (d4,)
    s["mot_embedding.2.blk.gcn.conv.weight"] = (3 * d, d4, 1, 1)          # model.py:45, blocks.py:47-54
    s["mot_embedding.2.blk.gcn.conv.bias"] = (3 * d,)
    s["mot_embedding.2.blk.tcn.weight"] = (d, d, 5, 1)                    # blocks.py:112-118
    s["mot_embedding.2.blk.tcn.bias"] = (d,)
    s["mot_embedding.5.blk.gcn.conv.weight"] = (2 * d, d, 1, 1)           # model.py:48
    s["mot_embedding.5.blk.gcn.conv.bias"] = (2 * d,)
    s["mot_embedding.5.blk.tcn.weight"] = (d, d, 3, 1)
    s["mot_embedding.5.blk.tcn.bias"] = (d,)
    for name, depth, heads, dh, mlp, adain in (
        ("encoder", c["encoder_depth"], c["encoder_heads"], c["encoder_dim_head"], c["encoder_mlp_dim"], False),
        ("decoder", c["decoder_depth"], c["decoder_heads"], c["decoder_dim_head"], c["decoder_mlp_dim"], True),
    ):
        inner = heads * dh
        for l in range(depth):
            p = f"{name}.layers.{l}"
            if adain:                                                     # transformer.py:98-107
                s[f"{p}.0.style.2.weight"] = (2 * d, d)
                s[f"{p}.0.style.2.bias"] = (2 * d,)
                s[f"{p}.0.style.4.weight"] = (2 * d, 2 * d)
                s[f"{p}.0.style.4.bias"] = (2 * d,)
            s[f"{p}.1.to_q.1.weight"] = (inner, d)                        # transformer.py:54-56
            s[f"{p}.1.to_k.1.weight"] = (inner, d)
            s[f"{p}.1.to_v.weight"] = (inner, d)
            s[f"{p}.1.to_out.0.weight"] = (d, inner)                      # transformer.py:58-61
            s[f"{p}.1.to_out.0.bias"] = (d,)
            s[f"{p}.2.net.0.weight"] = (mlp, d)                           # transformer.py:26-32
            s[f"{p}.2.net.0.bias"] = (mlp,)
            s[f"{p}.2.net.3.weight"] = (d, mlp)
            s[f"{p}.2.net.3.bias"] = (d,)
    s["to_mot.1.blk.gcn.conv.weight"] = (2 * d, d, 1, 1)                  # model.py:73
    s["to_mot.1.blk.gcn.conv.bias"] = (2 * d,)
    s["to_mot.1.blk.tcn.weight"] = (d, d, 3, 1)
    s["to_mot.1.blk.tcn.bias"] = (d,)
    s["to_mot.4.blk.gcn.conv.weight"] = (3 * d4, d, 1, 1)                 # model.py:76
    s["to_mot.4.blk.gcn.conv.bias"] = (3 * d4,)
    s["to_mot.4.blk.tcn.weight"] = (d4, d4, 5, 1)
    s["to_mot.4.blk.tcn.bias"] = (d4,)
    s["to_mot.6.weight"] = (cin, d4, 1, 1)                                # model.py:78
    s["to_mot.6.bias"] = (cin,)
    return s


def buffer_arrays(layout: str = "mocha") -> "OrderedDict[str, np.ndarray]":
    """The registered (non-learnable) buffers of the state_dict, regenerated (row a14)."""
    k = skeleton_constants(layout)
    b = OrderedDict()
    b["mot_embedding.2.A_j"] = k.A_j          # model.py:116-117
    b["mot_embedding.3.weight"] = k.pool      # net/graph.py:461
    b["mot_embedding.5.A_b"] = k.A_b          # model.py:144-145
    b["to_mot.1.A_b"] = k.A_b
    b["to_mot.3.weight"] = k.unpool           # net/graph.py:604
    b["to_mot.4.A_j"] = k.A_j
    return b


def synthetic_state_dict(seed: int = 1777, gain: float = 1.0, layout: str = "mocha",
                         cfg=None, with_buffers: bool = True) -> "OrderedDict[str, np.ndarray]":
    """Deterministic fp32 weights with the reference's names and shapes.

    Weights ~ U(-a, a) with a = gain / sqrt(fan_in) (the scale torch's default
    conv/linear init gives, so activations have trained-network-like magnitudes at
    gain≈1 and are larger at gain>1); biases ~ U(-a, a); pos_emb ~ N(0, 1) (model.py:40).
    """
    V = len(skeleton_constants(layout).parents)
    rng = np.random.Generator(np.random.PCG64(seed))
    out = OrderedDict()
    shapes = param_shapes(cfg, V)
    fan = {}
    for name, shp in shapes.items():
        if name == "pos_emb":
            out[name] = rng.standard_normal(shp).astype(np.float32)
            continue
        if name.endswith("weight"):
            fan_in = int(np.prod(shp[1:]))
            fan[name[:-len("weight")]] = fan_in
        else:
            fan_in = fan[name[:-len("bias")]]
        a = gain / np.sqrt(fan_in)
        out[name] = rng.uniform(-a, a, size=shp).astype(np.float32)
    if with_buffers:
        out.update(buffer_arrays(layout))
    return out


# --------------------------------------------------------------------------- CVAE
CVAE_CFG = dict(output_seq=90, latent_dim=256, depth=2, nheads=4, feedforward_dim=512)   # test_fullframework.py:52-55


def cvae_param_shapes(cfg=None) -> "OrderedDict[str, tuple]":
    """Learnable parameters of the reference ``CVAE`` that ``sample`` touches (model_CVAE.py:44-46):
    ``prior_net`` (:49-92) and ``decoder`` (:138-165).  The posterior ``encoder.*`` entries of a
    checkpoint are training-only and ignored; the ``pos_encoder.pe`` buffers are regenerated."""
    c = dict(CVAE_CFG, **(cfg or {}))
    d, ff = c["latent_dim"], c["feedforward_dim"]
    s = OrderedDict()
    s["prior_net.mu_token"] = (1, 1, d)
    s["prior_net.logvar_token"] = (1, 1, d)

    def attn(p):
        s[f"{p}.in_proj_weight"] = (3 * d, d)
        s[f"{p}.in_proj_bias"] = (3 * d,)
        s[f"{p}.out_proj.weight"] = (d, d)
        s[f"{p}.out_proj.bias"] = (d,)

    def ffn(p, norms):
        s[f"{p}.linear1.weight"] = (ff, d)
        s[f"{p}.linear1.bias"] = (ff,)
        s[f"{p}.linear2.weight"] = (d, ff)
        s[f"{p}.linear2.bias"] = (d,)
        for n in norms:
            s[f"{p}.{n}.weight"] = (d,)
            s[f"{p}.{n}.bias"] = (d,)

    for l in range(c["depth"]):
        p = f"prior_net.encoder.layers.{l}"
        attn(f"{p}.self_attn")
        ffn(p, ("norm1", "norm2"))
    for l in range(c["depth"]):
        p = f"decoder.decoder.layers.{l}"
        attn(f"{p}.self_attn")
        attn(f"{p}.multihead_attn")
        ffn(p, ("norm1", "norm2", "norm3"))
    return s


def sincos_pe(n: int, d: int = 256) -> np.ndarray:
    """``PositionalEncoding.pe[0, :n]`` of model_CVAE.py:168-178 (float32 arithmetic as in torch)."""
    import torch
    position = torch.arange(n).unsqueeze(1)
    div_term = torch.exp(torch.arange(0, d, 2) * (-np.log(10000.0) / d))
    pe = torch.zeros(n, d)
    pe[:, 0::2] = torch.sin(position * div_term)
    pe[:, 1::2] = torch.cos(position * div_term)
    return pe.numpy()


def synthetic_cvae_state_dict(seed: int = 99, gain: float = 1.0, cfg=None) -> "OrderedDict[str, np.ndarray]":
    """Deterministic fp32 CVAE weights with the reference's names and shapes."""
    rng = np.random.Generator(np.random.PCG64(seed))
    out = OrderedDict()
    for name, shp in cvae_param_shapes(cfg).items():
        if name.endswith("_token"):
            out[name] = rng.standard_normal(shp).astype(np.float32)
        elif ".norm" in name:
            base = 1.0 if name.endswith("weight") else 0.0
            out[name] = (base + 0.1 * rng.standard_normal(shp)).astype(np.float32)
        elif name.endswith("weight"):
            a = gain / np.sqrt(shp[1])
            out[name] = rng.uniform(-a, a, size=shp).astype(np.float32)
        else:
            out[name] = rng.uniform(-0.05, 0.05, size=shp).astype(np.float32)
    return out
