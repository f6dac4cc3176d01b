"""Host-side mirror of the reference ``Generator`` call surface on top of libmocha_hip.so.

The reference's boundary for this path is the attribute surface of its ``Generator``
module (model.py:15-106) as the demo uses it (test_fullframework.py:190-193, 301-302,
455-456, 465-467) plus ``mean_variance_norm`` (net/transformer.py:13-20) and the
``BallTree`` query (test_fullframework.py:294-296, 443).  This module offers the same
names with the same tensor-in/tensor-out meaning; PyTorch is used for device memory and
streams only — every value is computed by the HIP kernels behind the C ABI.
"""
from __future__ import annotations

import ctypes as C
from typing import Mapping, Optional

import numpy as np
import torch

from . import _C
from .skeleton import LAYOUT_ID, skeleton_constants
from .weights import DEFAULT_CFG, buffer_arrays, param_shapes

NTOK, DIM = 90, 256


def _ptr(t: Optional[torch.Tensor]):
    return C.c_void_p(0 if t is None else t.data_ptr())


class _CurrentStream:
    """Placeholder argument: ``_Context.call`` replaces it by the current stream of the CONTEXT's device (a caller whose
    current device differs from the model's would otherwise hand over a stream of the wrong device)."""


_STREAM = _CurrentStream()


def _stream():
    return _STREAM


def _dev_f32(t, device, shape_tail=None, name="tensor") -> torch.Tensor:
    if isinstance(t, np.ndarray):
        t = torch.from_numpy(t)
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name}: expected a tensor, got {type(t)}")
    if t.dtype != torch.float32:
        raise TypeError(f"{name}: expected float32, got {t.dtype}")
    t = t.to(device)
    if not t.is_contiguous():
        t = t.contiguous()
    if shape_tail is not None and tuple(t.shape[-len(shape_tail):]) != tuple(shape_tail):
        raise ValueError(f"{name}: trailing shape {tuple(t.shape)} does not end with {tuple(shape_tail)}")
    return t


class _Context:
    """Owns one ``mocha_ctx*``."""

    def __init__(self, cfg: Mapping, layout: str, device: torch.device):
        if device.type != "cuda":
            raise RuntimeError("the MOCHA HIP path needs a ROCm GPU device ('cuda:N'); there is no CPU fallback")
        self.lib = _C.load_library()
        self.device = device
        sk = skeleton_constants(layout)
        c = _C.mocha_cfg(
            T=cfg["nframes"], V=sk.V, C_in=cfg["mot_in_dim"], patch=cfg["temporal_patch_size"], dim=cfg["encoder_dim"],
            enc_depth=cfg["encoder_depth"], enc_heads=cfg["encoder_heads"], enc_dim_head=cfg["encoder_dim_head"],
            enc_mlp=cfg["encoder_mlp_dim"], dec_depth=cfg["decoder_depth"], dec_heads=cfg["decoder_heads"],
            dec_dim_head=cfg["decoder_dim_head"], dec_mlp=cfg["decoder_mlp_dim"], layout=LAYOUT_ID[layout])
        if cfg["decoder_dim"] != cfg["encoder_dim"]:
            raise ValueError("decoder_dim must equal encoder_dim")
        h = C.c_void_p()
        rc = self.lib.mocha_create(C.byref(c), device.index or 0, C.byref(h))
        _C.check(self.lib, None, rc, "mocha_create")
        self.h = h
        self.V = sk.V
        self.C_in = cfg["mot_in_dim"]
        self.T = cfg["nframes"]

    def call(self, fn: str, *args):
        with torch.cuda.device(self.device):
            if any(a is _STREAM for a in args):
                st = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
                args = tuple(st if a is _STREAM else a for a in args)
            rc = getattr(self.lib, fn)(self.h, *args)
        _C.check(self.lib, self.h, rc, fn)

    def generation(self) -> int:
        """Bumped by the library whenever it replaces a device buffer a captured graph may hold, or the bank changes."""
        return int(self.lib.mocha_generation(self.h))

    def __del__(self):
        try:
            if getattr(self, "h", None):
                self.lib.mocha_destroy(self.h)
                self.h = None
        except Exception:
            pass


_util_ctx = {}


def _utility_context(device: torch.device) -> _Context:
    """Weight-less context for the free function ``mean_variance_norm``."""
    key = (device.type, device.index)
    if key not in _util_ctx:
        _util_ctx[key] = _Context(DEFAULT_CFG, "mocha", device)
    return _util_ctx[key]


def mean_variance_norm(input: torch.Tensor, eps: float = 1e-5) -> torch.Tensor:
    """``mean_variance_norm`` of net/transformer.py:13-20 for the path's use: input (B, C=256, S)
    normalised over S per (b, c) with the unbiased std and eps outside the sqrt.  The demo
    passes ``encoded.permute(0, 2, 1)`` (test_fullframework.py:193); that view is consumed
    in place (no copy) and a (B, C, S) view of the result is returned."""
    if eps != 1e-5:
        raise ValueError("only eps=1e-5 (the reference default, the only value on the path) is supported")
    if input.dim() != 3 or input.shape[1] != DIM:
        raise ValueError(f"expected (B, {DIM}, S), got {tuple(input.shape)}")
    if not input.is_cuda:
        raise RuntimeError("mean_variance_norm: tensor must be on the GPU; there is no CPU fallback")
    tok_major = input.permute(0, 2, 1)             # (B, S, C)
    if not tok_major.is_contiguous():
        tok_major = tok_major.contiguous()
    if tok_major.dtype != torch.float32:
        raise TypeError("expected float32")
    B, S, _ = tok_major.shape
    if S != NTOK:
        raise ValueError(f"expected {NTOK} tokens, got {S}")
    ctx = _utility_context(input.device)
    out = torch.empty_like(tok_major)
    ctx.call("mocha_mvn", _ptr(tok_major), B, _ptr(out), _ptr(None), _ptr(None), _ptr(None), _stream())
    return out.permute(0, 2, 1)


class Generator:
    """Drop-in for the reference ``Generator`` (model.py:15) on one MI355X.

    >>> model = Generator(cfg['model']).load_state_dict(torch.load(path)['gen_ema']).eval()
    >>> tokens = model.mot_embedding(X); tokens = tokens + model.pos_emb[:, :tokens.shape[1]]
    >>> encoded = model.encoder(tokens); Y = model.to_mot(model.decoder(encoded, cha_encoded))
    """

    def __init__(self, config: Optional[Mapping] = None, layout: Optional[str] = None, device="cuda:0"):
        cfg = dict(DEFAULT_CFG)
        if config:
            cfg.update({k: v for k, v in config.items() if k in DEFAULT_CFG})
            if layout is None and "graph" in config:
                layout = config["graph"]["joint"].get("layout", "mocha")   # configs/config.yaml:42
        self.layout = layout or "mocha"
        self.cfg = cfg
        self.device = torch.device(device)
        self._ctx = _Context(cfg, self.layout, self.device)
        self.V = self._ctx.V
        self.pos_emb = None
        self._loaded = False

    # ---- nn.Module-like surface -------------------------------------------------------
    def eval(self):                                  # test_fullframework.py:49
        return self

    def to(self, device):
        if torch.device(device) != self.device:
            raise RuntimeError("create the Generator on its target device")
        return self

    def state_dict_keys(self):
        return list(param_shapes(self.cfg, self.V)) + list(buffer_arrays(self.layout))

    def load_state_dict(self, state_dict: Mapping, strict: bool = True):
        """Accepts the ``gen_ema`` mapping of trainer.py:239-240 (name -> tensor / ndarray).
        ``module.``-prefixed keys (nn.DataParallel checkpoints, trainer.py:45-47) are accepted."""
        lib, ctx = self._ctx.lib, self._ctx
        want = param_shapes(self.cfg, self.V)
        seen = set()
        for k, v in state_dict.items():
            name = k[len("module."):] if k.startswith("module.") else k
            a = v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)
            a = np.ascontiguousarray(a, dtype=np.float32)
            if name not in want and name not in buffer_arrays(self.layout):
                if strict:
                    raise KeyError(f"unexpected key in state_dict: {k}")
                continue
            shape = (C.c_int64 * a.ndim)(*a.shape)
            ctx.call("mocha_load_weight", name.encode(), a.ctypes.data_as(C.c_void_p), shape, a.ndim)
            seen.add(name)
            if name == "pos_emb":
                self.pos_emb = torch.from_numpy(a.copy()).to(self.device)
        missing = [k for k in want if k not in seen]
        if missing and strict:
            raise KeyError(f"missing keys in state_dict: {missing[:4]}{'...' if len(missing) > 4 else ''}")
        ctx.call("mocha_finalize_weights")
        self._loaded = True
        return self

    def reserve(self, max_batch: int):
        """Pre-allocate workspaces for chunks of ``max_batch`` windows."""
        self._ctx.call("mocha_reserve", int(max_batch))
        self._reserved = int(max_batch)
        return self

    def set_option(self, name: str, value: int):
        """Runtime options of the context, e.g. ``set_option("dual_stream", 1)`` (see include/mocha_hip.h)."""
        self._ctx.call("mocha_set_option", name.encode(), int(value))
        return self

    def linear(self, x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor] = None, engine: int = 0) -> torch.Tensor:
        """``torch.nn.functional.linear(x, weight, bias)`` for 2-D fp32 device tensors through one of the library's GEMM engines
        (``mocha_linear``): 0 = the kernel the path would pick, 1 = exact-f32 MFMA, 2 = bf16 x 3 planes."""
        x = _dev_f32(x, self.device, name="x")
        w = _dev_f32(weight, self.device, name="weight")
        if x.dim() != 2 or w.dim() != 2 or x.shape[1] != w.shape[1]:
            raise ValueError(f"linear: x {tuple(x.shape)} and weight {tuple(w.shape)} do not match")
        b = None if bias is None else _dev_f32(bias, self.device, (w.shape[0],), "bias")
        y = torch.empty((x.shape[0], w.shape[0]), dtype=torch.float32, device=self.device)
        self._ctx.call("mocha_linear", _ptr(x), _ptr(w), _ptr(b), _ptr(y), int(x.shape[0]), int(w.shape[0]), int(x.shape[1]), int(engine),
                       _stream())
        return y

    # ---- measurement support -----------------------------------------------------------
    def profile_start(self):
        """Bracket every kernel launch with HIP events until ``profile_stop`` (bench.py)."""
        self._ctx.call("mocha_profile_start")

    def profile_stop(self) -> dict:
        import json
        buf = C.create_string_buffer(1 << 16)
        self._ctx.call("mocha_profile_stop", buf, len(buf))
        return json.loads(buf.value.decode())

    # ---- sub-module call surface of the demo -----------------------------------------
    def _need(self):
        if not self._loaded:
            raise RuntimeError("weights not loaded: call load_state_dict first")

    def _x(self, X, name):
        return _dev_f32(X, self.device, (self.cfg["nframes"], self.V, self.cfg["mot_in_dim"]), name)

    def _tok(self, t, name):
        return _dev_f32(t, self.device, (NTOK, DIM), name)

    def mot_embedding(self, X):                      # model.py:42-50
        self._need()
        X = self._x(X, "X")
        out = torch.empty((X.shape[0], NTOK, DIM), dtype=torch.float32, device=self.device)
        self._ctx.call("mocha_embed", _ptr(X), X.shape[0], _ptr(out), 0, _stream())
        return out

    def encoder(self, tokens):                       # model.py:53-59
        self._need()
        tokens = self._tok(tokens, "tokens")
        out = torch.empty_like(tokens)
        self._ctx.call("mocha_encoder", _ptr(tokens), tokens.shape[0], _ptr(out), _stream())
        return out

    def decoder(self, src_encoded, cha_encoded):     # model.py:62-68
        self._need()
        s = self._tok(src_encoded, "src_encoded")
        c = self._tok(cha_encoded, "cha_encoded")
        if s.shape[0] != c.shape[0]:
            raise ValueError("decoder: batch sizes differ")
        out = torch.empty_like(s)
        self._ctx.call("mocha_decoder", _ptr(s), _ptr(c), s.shape[0], _ptr(out), _stream())
        return out

    def style_constants(self, cha_encoded):
        """AdaIN's gamma / beta of every decoder layer for character features (net/transformer.py:98-107: token mean -> Linear ->
        LeakyReLU -> Linear), (B, 512 * decoder_depth) = [gamma_0 | beta_0 | gamma_1 | beta_1 ...] - what ``decoder`` derives from its
        second argument first, and what a bank caches per entry."""
        self._need()
        c = self._tok(cha_encoded, "cha_encoded")
        out = torch.empty((c.shape[0], 512 * int(self.cfg["decoder_depth"])), dtype=torch.float32, device=self.device)
        self._ctx.call("mocha_style_constants", _ptr(c), c.shape[0], _ptr(out), _stream())
        return out

    def to_mot(self, tokens):                        # model.py:71-80
        self._need()
        tokens = self._tok(tokens, "tokens")
        B = tokens.shape[0]
        out = torch.empty((B, self.cfg["nframes"], self.V, self.cfg["mot_in_dim"]), dtype=torch.float32, device=self.device)
        self._ctx.call("mocha_to_mot", _ptr(tokens), B, _ptr(out), _stream())
        return out

    def forward(self, src_X, cha_X, extract_feature: bool = False):     # model.py:82-106
        self._need()
        s, c = self._x(src_X, "src_X"), self._x(cha_X, "cha_X")
        if s.shape[0] != c.shape[0]:
            raise ValueError("forward: batch sizes differ")
        B = s.shape[0]
        if extract_feature:
            outs = [torch.empty((B, NTOK, DIM), dtype=torch.float32, device=self.device) for _ in range(4)]
            self._ctx.call("mocha_forward_features", _ptr(s), _ptr(c), B, *[_ptr(o) for o in outs], _stream())
            return tuple(outs)                       # src_encoded, cha_encoded, src_cnt, cha_cnt
        Y = torch.empty_like(s)
        self._ctx.call("mocha_forward", _ptr(s), _ptr(c), B, _ptr(Y), _stream())
        return Y

    __call__ = forward

    # ---- fused extras (same arithmetic, fewer launches) --------------------------------
    def set_pose_norm(self, X_mean, X_std, Y_mean, Y_std):
        """The demo's norm.npz arrays (test_fullframework.py:64-71), any shape ending in (V+1, C) with the
        root bone first.  Enables ``raw=True``: un-normalised poses with the root bone go in
        ((B,T,V+1,C)), the z-score of :186 and the de-normalisation of :303 run inside the kernels."""
        n = (self.V + 1) * self.cfg["mot_in_dim"]
        arrs = []
        for a in (X_mean, X_std, Y_mean, Y_std):
            a = np.ascontiguousarray(np.asarray(a, dtype=np.float32).reshape(-1))
            if a.size != n:
                raise ValueError(f"pose norm arrays must hold (V+1)*C = {n} values, got {a.size}")
            arrs.append(a)
        self._ctx.call("mocha_set_pose_norm", *[a.ctypes.data_as(C.c_void_p) for a in arrs])
        self._has_pose_norm = True
        return self

    def featurize(self, Yrot, Ypos, Yvel, Yang):
        """The demo's featurisation on the device (test_fullframework.py:141-185): local bone features of B
        windows, root bone first — Yrot (B,T,V+1,4) (w,x,y,z), Ypos/Yvel/Yang (B,T,V+1,3) — to the
        un-normalised X (B,T,V+1,15) that ``encode(..., raw=True)`` / ``characterize(..., raw=True)`` take."""
        T, J = self.cfg["nframes"], self.V + 1
        r = _dev_f32(Yrot, self.device, (T, J, 4), "Yrot")
        ps = [_dev_f32(a, self.device, (T, J, 3), n) for a, n in ((Ypos, "Ypos"), (Yvel, "Yvel"), (Yang, "Yang"))]
        B = r.shape[0]
        if any(p.shape[0] != B for p in ps):
            raise ValueError("featurize: batch sizes differ")
        X = torch.empty((B, T, J, self.cfg["mot_in_dim"]), dtype=torch.float32, device=self.device)
        self._ctx.call("mocha_featurize", _ptr(r), _ptr(ps[0]), _ptr(ps[1]), _ptr(ps[2]), B, _ptr(X), _stream())
        return X

    def characterize_pair(self, src_X, cha_X, cnt_mean, cnt_std, return_index: bool = False, return_bank: bool = False,
                          raw: bool = False):
        """The demo pair in one pass: the character clip becomes the bank and the source clip is characterized against
        it — ``encode(cha)`` + ``ContextBank`` + ``characterize(src)`` with both clips sharing every launch of
        mot_embedding / encoder / cnt (test_fullframework.py:188-194, 271-277, 293-296, 438-443, 465-467).  The bank is
        transient; ``return_bank`` also returns (cha_encoded, cha_cnt_nm) for a later ``ContextBank``.
        Result order: Y[, idx][, cha_encoded, cha_cnt_nm].
        ``cnt_mean`` / ``cnt_std`` may be NumPy arrays (as ``np.load('cnt_norm.npz')`` gives them), but then every call copies them to
        the device - a pageable host copy waits for the stream to drain (measured: 138 µs of idle GPU per call at 585 + 585 windows);
        in a loop pass device tensors, converted once."""
        self._need()
        conv = (lambda X, n: self._xraw(X, n)) if raw else (lambda X, n: self._x(X, n))
        s, c = conv(src_X, "src_X"), conv(cha_X, "cha_X")
        mean = _dev_f32(cnt_mean, self.device, (NTOK, DIM), "cnt_mean")
        std = _dev_f32(cnt_std, self.device, (NTOK, DIM), "cnt_std")
        Bs, Bc = s.shape[0], c.shape[0]
        Y = torch.empty((Bs, self.cfg["nframes"], self.V, self.cfg["mot_in_dim"]), dtype=torch.float32, device=self.device)
        idx = torch.empty((Bs,), dtype=torch.int32, device=self.device)
        enc = nm = None
        if return_bank:
            enc = torch.empty((Bc, NTOK, DIM), dtype=torch.float32, device=self.device)
            nm = torch.empty_like(enc)
        self._ctx.call("mocha_characterize_pair_raw" if raw else "mocha_characterize_pair", _ptr(s), Bs, _ptr(c), Bc, _ptr(mean), _ptr(std),
                       _ptr(Y), _ptr(idx), _ptr(enc) if return_bank else None, _ptr(nm) if return_bank else None, _stream())
        out = (Y,) + ((idx,) if return_index else ()) + ((enc, nm) if return_bank else ())
        return out[0] if len(out) == 1 else out

    def _xraw(self, X, name):
        return _dev_f32(X, self.device, (self.cfg["nframes"], self.V + 1, self.cfg["mot_in_dim"]), name)

    def encode(self, X, cnt_mean=None, cnt_std=None, raw: bool = False):
        """The demo's encode sequence (test_fullframework.py:190-193) in one call.
        Returns (encoded, cnt) or (encoded, cnt, cnt_nm) when the global cnt norm is given
        (cnt_nm = (cnt - cnt_mean) / cnt_std, test_fullframework.py:293,442)."""
        self._need()
        X = self._xraw(X, "X_raw") if raw else self._x(X, "X")
        fn = "mocha_encode_raw" if raw else "mocha_encode"
        B = X.shape[0]
        enc = torch.empty((B, NTOK, DIM), dtype=torch.float32, device=self.device)
        cnt = torch.empty_like(enc)
        if cnt_mean is None:
            self._ctx.call(fn, _ptr(X), B, _ptr(enc), _ptr(cnt), _ptr(None), _ptr(None), _ptr(None), _stream())
            return enc, cnt
        m = _dev_f32(cnt_mean, self.device, (NTOK, DIM), "cnt_mean")
        sd = _dev_f32(cnt_std, self.device, (NTOK, DIM), "cnt_std")
        nm = torch.empty_like(enc)
        self._ctx.call(fn, _ptr(X), B, _ptr(enc), _ptr(cnt), _ptr(m), _ptr(sd), _ptr(nm), _stream())
        return enc, cnt, nm


class ContextBank:
    """The character feature bank with the demo's BallTree role (test_fullframework.py:293-298):
    ``ContextBank(model, cha_cnt_nm, cha_encoded).query(src_cnt_nm)`` -> (dist, idx) like
    ``BallTree.query(k=1)``; ``gather(idx)`` -> ``cha_encoded[idx]``."""

    def __init__(self, model: Generator, cha_cnt_nm, cha_encoded, copy: bool = False, bf16: bool = False, dec_cache: bool = True):
        model._need()
        self.model = model
        self._dec_cache = dec_cache  # False: no per-entry decoder constants at mocha_bank_set (+ 92 KB per entry) - for banks whose decode never reads them
        dev = model.device
        self.cnt_nm = _dev_f32(cha_cnt_nm, dev, None, "cha_cnt_nm").reshape(-1, NTOK * DIM)
        self.encoded = _dev_f32(cha_encoded, dev, (NTOK, DIM), "cha_encoded")
        if self.cnt_nm.shape[0] != self.encoded.shape[0]:
            raise ValueError("bank: cnt_nm and encoded have different entry counts")
        self.N = self.cnt_nm.shape[0]
        self._copy = copy
        self._bf16 = bf16          # match against a bf16 copy of cnt_nm (half the bytes per bank scan)
        self.activate()

    @classmethod
    def received(cls, model: "Generator", n_entries: int, bf16: bool = False) -> "ContextBank":
        """Handle for a bank that ``mocha_bank_broadcast`` delivered into the context's own buffers on a non-root rank
        (distributed.bank_broadcast): it has no tensors of its own and is the context's current bank."""
        self = cls.__new__(cls)
        self.model, self.N, self._copy, self._bf16 = model, int(n_entries), True, bf16
        self.cnt_nm = self.encoded = None
        model._bank = self
        return self

    def tensors(self):
        """(cnt_nm (N, 23040), encoded (N, 90, 256)) as device tensors: the bank's own, or - for a bank that arrived by
        ``mocha_bank_broadcast`` and lives in the context - fresh copies (``mocha_bank_export``) on every call."""
        if self.cnt_nm is None:
            if getattr(self.model, "_bank", None) is not self:
                raise RuntimeError("a received bank lives in the context and was replaced by another bank: broadcast it again")
            nm = torch.empty((self.N, NTOK * DIM), dtype=torch.float32, device=self.model.device)
            enc = torch.empty((self.N, NTOK, DIM), dtype=torch.float32, device=self.model.device)
            self.model._ctx.call("mocha_bank_export", _ptr(nm), _ptr(enc), _stream())
            return nm, enc
        return self.cnt_nm, self.encoded

    def activate(self):
        """Make this bank the context's current bank (borrowed buffers unless copy=True)."""
        if self.cnt_nm is None:
            if getattr(self.model, "_bank", None) is not self:
                raise RuntimeError("a received bank lives in the context and was replaced by another bank: broadcast it again")
            return self
        flags = (0 if self._copy else 1) | (2 if self._bf16 else 0)
        if not getattr(self, "_dec_cache", True):
            self.model.set_option("bank_dec_cache", 0)
        try:
            self.model._ctx.call("mocha_bank_set", _ptr(self.cnt_nm), _ptr(self.encoded), self.N, flags, _stream())
        finally:
            if not getattr(self, "_dec_cache", True):
                self.model.set_option("bank_dec_cache", 1)
        self.model._bank = self
        return self

    def query(self, query_nm, k: int = 1, return_distance: bool = True):
        """``BallTree.query``: (dist (Q,k), idx (Q,k)), distances ascending.  k = 1 is the path (test_fullframework.py:296,443)
        and takes the fast matchers; k > 1 is an exact k-pass selection over every row's distance (``mocha_match_topk``)."""
        if getattr(self.model, "_bank", None) is not self:
            self.activate()
        q = _dev_f32(query_nm, self.model.device, None, "query").reshape(-1, NTOK * DIM)
        Q = q.shape[0]
        if k != 1:
            if not 1 < k <= 64:
                raise ValueError("k must be in 1 .. 64")
            idx = torch.empty((Q, k), dtype=torch.int32, device=q.device)
            dist = torch.empty((Q, k), dtype=torch.float32, device=q.device) if return_distance else None
            self.model._ctx.call("mocha_match_topk", _ptr(q), Q, int(k), _ptr(idx), _ptr(dist), _stream())
            return (dist, idx) if return_distance else idx
        idx = torch.empty((Q,), dtype=torch.int32, device=q.device)
        dist = torch.empty((Q,), dtype=torch.float32, device=q.device) if return_distance else None
        self.model._ctx.call("mocha_match", _ptr(q), Q, _ptr(idx), _ptr(dist), _stream())
        if return_distance:
            return dist[:, None], idx[:, None]
        return idx[:, None]

    def gather_blend(self, idx, dist, temperature: float = 1.0):
        """Soft context matching over the k neighbours of ``query(q, k)``: softmax(-dist / temperature)-weighted sum of their
        ``encoded`` entries, (Q,90,256).  An extension (SURVEY.md §8f N4 "optional"): the reference gathers the single nearest."""
        if getattr(self.model, "_bank", None) is not self:
            self.activate()
        idx = idx.to(device=self.model.device, dtype=torch.int32).contiguous()
        d = dist.to(self.model.device, torch.float32).contiguous()
        Q, k = idx.shape
        out = torch.empty((Q, NTOK, DIM), dtype=torch.float32, device=self.model.device)
        self.model._ctx.call("mocha_bank_gather_blend", _ptr(idx), _ptr(d), C.c_float(temperature), Q, k, _ptr(out), _stream())
        return out

    def gather(self, idx):
        if getattr(self.model, "_bank", None) is not self:
            self.activate()
        idx = idx.reshape(-1).to(device=self.model.device, dtype=torch.int32).contiguous()
        out = torch.empty((idx.shape[0], NTOK, DIM), dtype=torch.float32, device=self.model.device)
        self.model._ctx.call("mocha_bank_gather", _ptr(idx), idx.shape[0], _ptr(out), _stream())
        return out

    def characterize(self, src_X, cnt_mean, cnt_std, return_index: bool = False, raw: bool = False):
        """NN ('cm_') branch of the demo for all source windows at once: encode, z-score,
        1-NN match, gather, decode, to_mot (test_fullframework.py:188-194,438-443,465-467).
        raw=True (after Generator.set_pose_norm): src_X is un-normalised with the root bone,
        (B,T,V+1,C); the result is de-normalised (B,T,V,C)."""
        if getattr(self.model, "_bank", None) is not self:
            self.activate()
        m = self.model
        X = m._xraw(src_X, "src_X_raw") if raw else m._x(src_X, "src_X")
        mean = _dev_f32(cnt_mean, m.device, (NTOK, DIM), "cnt_mean")
        std = _dev_f32(cnt_std, m.device, (NTOK, DIM), "cnt_std")
        Y = torch.empty((X.shape[0], m.cfg["nframes"], m.V, m.cfg["mot_in_dim"]), dtype=torch.float32, device=m.device)
        idx = torch.empty((X.shape[0],), dtype=torch.int32, device=m.device)
        m._ctx.call("mocha_characterize_raw" if raw else "mocha_characterize", _ptr(X), X.shape[0], _ptr(mean), _ptr(std),
                    _ptr(Y), _ptr(idx), _stream())
        return (Y, idx) if return_index else Y


class StreamingCharacterizer:
    """Window-by-window characterization (BASELINE configs[4]: a clip streamed one 60-frame window
    per step against a large bank).  The whole per-window step — encode, z-score, 1-NN bank scan,
    gather, decoder, to_mot (test_fullframework.py:438-443,465-467) — is captured once into a HIP
    graph by the library (``mocha_step_graph``) and replayed per window, so the ~45 kernel launches cost one
    graph launch.  The library re-captures by itself when the bank or a workspace it baked in has changed; this class
    re-activates its bank when another bank was made current in between."""

    def __init__(self, bank: ContextBank, cnt_mean, cnt_std, use_graph: bool = True, lanes: int = 1, raw: bool = False):
        """raw=True (after ``Generator.set_pose_norm``): windows come un-normalised with the root bone, (60, V+1, 15), as
        ``Generator.featurize`` writes them; the z-score of test_fullframework.py:186 and the de-normalisation of :303 run inside the
        step's kernels, the result is (60, V, 15)."""
        self.bank, self.model = bank, bank.model
        m = self.model
        self.raw = bool(raw)
        if not 1 <= lanes <= 3:
            raise ValueError("lanes must be 1..3")
        if lanes > 1 and not use_graph:
            raise ValueError("lanes > 1 needs the captured step (use_graph=True)")
        self.lanes = lanes
        self.mean = _dev_f32(cnt_mean, m.device, (NTOK, DIM), "cnt_mean")
        self.std = _dev_f32(cnt_std, m.device, (NTOK, DIM), "cnt_std")
        shape = (1, m.cfg["nframes"], m.V, m.cfg["mot_in_dim"])
        in_shape = (1, m.cfg["nframes"], m.V + 1, m.cfg["mot_in_dim"]) if self.raw else shape
        self.xs = [torch.zeros(in_shape, dtype=torch.float32, device=m.device) for _ in range(lanes)]
        self.ys = [torch.empty(shape, dtype=torch.float32, device=m.device) for _ in range(lanes)]
        self.idxs = [torch.zeros((1,), dtype=torch.int32, device=m.device) for _ in range(lanes)]
        self.x, self.y, self.idx = self.xs[0], self.ys[0], self.idxs[0]
        self.use_graph = use_graph
        self._lane_streams = None
        if lanes > 1:
            m.set_option("lanes", lanes)                   # re-plans the workspace sets (generation moves)
        bank.activate()

    def _enqueue(self, lane: int = 0, check_bank: bool = True):
        if check_bank and getattr(self.model, "_bank", None) is not self.bank:
            self.bank.activate()                           # another bank was made current: ours again (bumps the generation)
        if self.use_graph:
            self.model._ctx.call("mocha_step_graph_lane", lane, _ptr(self.xs[lane]), _ptr(self.mean), _ptr(self.std), _ptr(self.ys[lane]),
                                 _ptr(self.idxs[lane]), 1 if self.raw else 0, _stream())
        else:
            self.model._ctx.call("mocha_characterize_raw" if self.raw else "mocha_characterize", _ptr(self.x), 1, _ptr(self.mean), _ptr(self.std), _ptr(self.y),
                                 _ptr(self.idx), _stream())

    @property
    def input(self) -> torch.Tensor:
        """The captured step's own input window (1, 60, V, 15): a producer that writes the next window straight into it - the
        featurisation kernel, a sensor copy - and then calls ``step()`` without an argument saves the device-to-device copy
        (one 4 µs kernel of the 0.41 ms step)."""
        return self.x

    def step(self, window: Optional[torch.Tensor] = None):
        """window (60, V, 15) or (1, 60, V, 15) on the GPU -> (Y (60, V, 15) view, idx tensor view); both
        are overwritten by the next step.  Without an argument the window already in ``self.input`` is characterized."""
        if window is not None:
            self.x.copy_(window.reshape(self.x.shape), non_blocking=True)
        self._enqueue()
        return self.y[0], self.idx

    def run_clip(self, windows: torch.Tensor):
        """All W windows of a clip, one captured per-window step each, with up to ``lanes`` steps in flight: window i runs on
        lane i % lanes and that lane's stream, so window i + 1's encode chain overlaps window i's bank scan (the windows of a
        clip are known up front and independent: test_fullframework.py:128, 148-158, 438).  Same kernels and the same
        arithmetic per window as ``step``: results are bit-identical to it.  -> (Y (W,60,V,15), idx (W,)), ordered on the
        caller's stream when the call returns (no host synchronisation)."""
        m = self.model
        wins = _dev_f32(windows, m.device, tuple(self.x.shape[1:]), "windows")
        W = wins.shape[0]
        Y = torch.empty((W,) + tuple(self.y.shape[1:]), dtype=torch.float32, device=m.device)
        idx = torch.empty((W,), dtype=torch.int32, device=m.device)
        if W == 0:
            return Y, idx
        if getattr(m, "_bank", None) is not self.bank:
            self.bank.activate()                           # once, on the caller's stream, before the lanes fork
        if self._lane_streams is None:
            self._lane_streams = [torch.cuda.Stream(device=m.device) for _ in range(self.lanes)]
        cur = torch.cuda.current_stream(m.device)
        ready = torch.cuda.Event(); ready.record(cur)
        for st in self._lane_streams:
            st.wait_event(ready)                           # inputs (and the outputs' allocations) are ordered before the lanes start
        for i in range(W):
            k = i % self.lanes
            with torch.cuda.stream(self._lane_streams[k]):
                self.xs[k].copy_(wins[i:i + 1], non_blocking=True)
                self._enqueue(k, check_bank=False)
                Y[i:i + 1].copy_(self.ys[k], non_blocking=True)
                idx[i:i + 1].copy_(self.idxs[k], non_blocking=True)
        for st in self._lane_streams:
            done = torch.cuda.Event(); done.record(st)
            cur.wait_event(done)
        for t in (wins, Y, idx):
            for st in self._lane_streams:
                t.record_stream(st)
        return Y, idx


class CVAE:
    """Drop-in for the reference ``CVAE`` sampler (model_CVAE.py:8-46) as the demo uses it
    (test_fullframework.py:52-58, 446-449): ``CVAE(...).load_state_dict(sd).eval().sample(condition)``.
    Only the inference path (``sample``, ``prior``) is provided; the posterior encoder is training-only."""

    def __init__(self, output_seq: int = 90, latent_dim: int = 256, depth: int = 2, nheads: int = 4,
                 feedforward_dim: int = 512, dropout: float = 0.1, activation=None, device="cuda:0"):
        if (output_seq, latent_dim, depth, nheads, feedforward_dim) != (90, 256, 2, 4, 512):
            raise ValueError("the HIP CVAE is built for the demo's configuration (90, 256, 2, 4, 512)")
        self.device = torch.device(device)
        self._ctx = _Context(DEFAULT_CFG, "mocha", self.device)
        self._loaded = False

    def eval(self):
        return self

    def to(self, device):
        if torch.device(device) != self.device:
            raise RuntimeError("create the CVAE on its target device")
        return self

    def load_state_dict(self, state_dict: Mapping, strict: bool = True):
        """The bare ``state_dict`` the demo loads (``torch.load('cvae_020000.pt')``, test_fullframework.py:56-58): tensors or
        ndarrays by the reference's names.  The training-only posterior ``encoder.*`` entries and the ``pos_encoder.pe`` buffers
        are accepted; ``module.``-prefixed keys (a DataParallel-wrapped save) too.  ``strict=False`` skips unknown keys and lets a
        re-load bring only some tensors (a context that never saw a tensor still fails, in ``mocha_cvae_finalize``)."""
        from .weights import cvae_param_shapes
        want = cvae_param_shapes()
        seen = set()
        has_pe = False
        for k0, v in state_dict.items():
            k = k0[len("module."):] if k0.startswith("module.") else k0
            a = v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)
            a = np.ascontiguousarray(a, dtype=np.float32)
            if k not in want and not k.startswith("encoder.") and not k.endswith("pos_encoder.pe"):
                if strict:
                    raise KeyError(f"unexpected key in CVAE state_dict: {k0}")
                continue
            shape = (C.c_int64 * a.ndim)(*a.shape)
            self._ctx.call("mocha_cvae_load_weight", k.encode(), a.ctypes.data_as(C.c_void_p), shape, a.ndim)
            seen.add(k)
            has_pe = has_pe or k == "prior_net.pos_encoder.pe"
        missing = [k for k in want if k not in seen]
        if missing and strict:
            raise KeyError(f"missing keys in CVAE state_dict: {missing[:4]}{'...' if len(missing) > 4 else ''}")
        if not has_pe:
            # regenerate the registered buffer exactly as torch builds it (model_CVAE.py:168-178)
            from .weights import sincos_pe
            pe = np.ascontiguousarray(sincos_pe(192)[None], dtype=np.float32)
            self._ctx.call("mocha_cvae_load_weight", b"prior_net.pos_encoder.pe", pe.ctypes.data_as(C.c_void_p),
                           (C.c_int64 * 3)(*pe.shape), 3)
        self._ctx.call("mocha_cvae_finalize")
        self._loaded = True
        return self

    def _run(self, c, eps):
        if not self._loaded:
            raise RuntimeError("CVAE weights not loaded: call load_state_dict first")
        c = _dev_f32(c, self.device, (180, DIM), "condition")
        B = c.shape[0]
        out = torch.empty((B, NTOK, DIM), dtype=torch.float32, device=self.device)
        mu = torch.empty((B, DIM), dtype=torch.float32, device=self.device)
        logvar = torch.empty_like(mu)
        e = None if eps is None else _dev_f32(eps, self.device, (DIM,), "eps")
        self._ctx.call("mocha_cvae_sample", _ptr(c), B, _ptr(out), _ptr(mu), _ptr(logvar), _ptr(e), _stream())
        return out, mu, logvar

    def prior(self, c):                                      # model_CVAE.py:32-34
        _, mu, logvar = self._run(c, None)
        return mu, logvar

    def sample(self, c, deterministic: bool = False, eps: Optional[torch.Tensor] = None):    # model_CVAE.py:44-46
        """``eps`` (B, 256) is the reparameterisation noise; when omitted and not deterministic it is drawn
        with ``torch.randn`` on the device (the reference draws ``torch.randn_like(std)``, :83)."""
        if deterministic:
            eps = None
        elif eps is None:
            eps = torch.randn((c.shape[0], DIM), dtype=torch.float32, device=self.device)
        return self._run(c, eps)[0]


class OursSession:
    """The demo's CVAE ("Ours") branch, frame by frame and entirely on the device
    (test_fullframework.py:446-457): condition = cat[z-scored src cnt, z-scored previous character feature]
    -> CVAE.sample -> de-normalise -> decoder -> to_mot, feeding the sampled feature back.  One clip, or B clips
    advanced in lock step (the branch is autoregressive in time; clips are its data-parallel axis).  The ~65 small
    kernels of a frame are captured into a HIP graph per sampling mode and replayed (``use_graph``)."""

    def __init__(self, model: Generator, cvae: "CVAE", src_cnt_mean, src_cnt_std, cha_encoded_mean, cha_encoded_std,
                 use_graph: bool = True):
        if cvae.device != model.device:
            raise ValueError("CVAE and Generator must be on the same device")
        self.model, self.cvae = model, cvae
        d = model.device
        self.sm = _dev_f32(src_cnt_mean, d, (NTOK, DIM), "src_cnt_mean")
        self.ss = _dev_f32(src_cnt_std, d, (NTOK, DIM), "src_cnt_std")
        self.cm = _dev_f32(cha_encoded_mean, d, (NTOK, DIM), "cha_encoded_mean")
        self.cs = _dev_f32(cha_encoded_std, d, (NTOK, DIM), "cha_encoded_std")
        self.use_graph = use_graph
        self.prev = None

    def reset(self, first_cha_encoded):
        """prev_cha_encoded = curr_cha_encoded.clone() of the first frame (test_fullframework.py:436).  One (90,256) feature
        for a single clip, or (B,90,256) for B clips."""
        m, d = self.model, self.model.device
        f = _dev_f32(first_cha_encoded, d, (NTOK, DIM), "cha_encoded").reshape(-1, NTOK, DIM)
        B = f.shape[0]
        if self.prev is None or self.prev.shape[0] != B:
            new = lambda *shape: torch.empty(shape, dtype=torch.float32, device=d)
            self.se, self.sc = new(B, NTOK, DIM), new(B, NTOK, DIM)
            self.cond, self.vae, self.cur, self.dec = new(B, 2 * NTOK, DIM), new(B, NTOK, DIM), new(B, NTOK, DIM), new(B, NTOK, DIM)
            # prev_cha_encoded IS the previous frame's curr_cha_encoded (test_fullframework.py:452): the condition kernel reads it before
            # this frame's sample overwrites it, so one buffer serves both (no copy per frame)
            self.prev = self.cur
            self.eps, self.mu, self.logvar = new(B, DIM), new(B, DIM), new(B, DIM)
            self.Y = new(B, m.cfg["nframes"], m.V, m.cfg["mot_in_dim"])
            self._graphs = {}
        self.prev.copy_(f)
        return self

    def _enqueue(self, with_eps: bool):
        m, c, B, st = self.model._ctx, self.cvae._ctx, self.prev.shape[0], _stream()
        c.call("mocha_cvae_condition", _ptr(self.sc), _ptr(self.sm), _ptr(self.ss), _ptr(self.prev), _ptr(self.cm), _ptr(self.cs),
               B, _ptr(self.cond), st)
        c.call("mocha_cvae_sample", _ptr(self.cond), B, _ptr(self.vae), _ptr(self.mu), _ptr(self.logvar),
               _ptr(self.eps) if with_eps else None, st)
        c.call("mocha_scale_shift", _ptr(self.vae), _ptr(self.cm), _ptr(self.cs), B, _ptr(self.cur), st)
        m.call("mocha_decoder", _ptr(self.se), _ptr(self.cur), B, _ptr(self.dec), st)
        m.call("mocha_to_mot", _ptr(self.dec), B, _ptr(self.Y), st)

    def step(self, src_encoded, src_cnt, eps=None, deterministic: bool = False):
        """One frame of every clip: returns (trans_Ytil (B,60,V,15), curr_cha_encoded (B,90,256)), B = 1 for (90,256) inputs.
        Both are views of session buffers that the next step overwrites.  ``eps`` (B,256) is the sampler's noise
        (drawn with torch.randn when omitted and not deterministic)."""
        if self.prev is None:
            raise RuntimeError("call reset(first_cha_encoded) first")
        d, B = self.model.device, self.prev.shape[0]
        se = _dev_f32(src_encoded, d, (NTOK, DIM), "src_encoded").reshape(-1, NTOK, DIM)
        sc = _dev_f32(src_cnt, d, (NTOK, DIM), "src_cnt").reshape(-1, NTOK, DIM)
        if se.shape[0] != B or sc.shape[0] != B:
            raise ValueError(f"OursSession.step: expected features of {B} clip(s), got {se.shape[0]} / {sc.shape[0]}")
        self.se.copy_(se, non_blocking=True); self.sc.copy_(sc, non_blocking=True)
        with_eps = not deterministic
        if with_eps:
            if eps is None:
                self.eps.normal_()
            else:
                self.eps.copy_(_dev_f32(eps, d, (DIM,), "eps").reshape(B, DIM), non_blocking=True)
        if not self.use_graph:
            self._enqueue(with_eps)
        else:
            gen = (self.model._ctx.generation(), self.cvae._ctx.generation())
            if getattr(self, "_graph_gen", None) != gen:          # a workspace a graph baked in was replaced: capture again
                self._graphs = {}
            g = self._graphs.get(with_eps)
            if g is None:
                keep = self.prev.clone()
                self._enqueue(with_eps)                       # warm-up outside capture (lazy allocations), then undo its state update
                torch.cuda.synchronize(d)
                self.prev.copy_(keep)
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    self._enqueue(with_eps)
                self._graphs[with_eps] = g
                self._graph_gen = (self.model._ctx.generation(), self.cvae._ctx.generation())
                self.prev.copy_(keep)
            g.replay()
        return self.Y, self.cur
