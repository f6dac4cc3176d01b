"""Character feature bank: build, statistics, on-disk format, and a bank-sharded matcher.

* ``build_bank``   — what collect_CVAE_feature_action.py:167-180 / compute_cnt_norm.py:157-175 do: batch-encode all
  60-frame windows of a character database into ``encoded`` and ``cnt`` and take the per-(token, channel) mean / std
  of ``cnt`` over the entries.  All arithmetic runs in the HIP kernels; this module only loops and stores.
* ``save_bank`` / ``load_bank`` — the reference's ``np.savez_compressed(encoded=, cnt=, range_starts=, range_stops=,
  action_label=)`` feature file (collect_CVAE_feature_action.py:185-189) and ``cnt_norm.npz`` (mean, std;
  compute_cnt_norm.py:178-179).
* ``ShardedContextBank`` — each rank scans its own block of bank rows and the per-query (distance, index) pairs are
  all-gathered (8 bytes per query and rank); every rank then holds the global winner.  Cuts the HBM-bound scan of a
  streamed query by the number of GPUs; the ``encoded`` features are replicated so the gather stays local.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import numpy as np
import torch

from . import distributed as D
from .generator import DIM, NTOK, ContextBank, Generator, _dev_f32, _ptr, _stream


def build_bank(model: Generator, X, batch: int = 1024, raw: bool = False):
    """X (N,60,V,15) z-scored windows (or un-normalised with the root bone when ``raw``) -> dict with device tensors
    ``encoded`` (N,90,256), ``cnt`` (N,90,256), ``cnt_mean`` / ``cnt_std`` (90,256)."""
    enc, cnt = [], []
    for s in range(0, len(X), batch):
        e, c = model.encode(X[s:s + batch], raw=raw)
        enc.append(e); cnt.append(c)
    encoded, cntf = torch.cat(enc), torch.cat(cnt)
    mean = torch.empty((NTOK, DIM), dtype=torch.float32, device=model.device)
    std = torch.empty_like(mean)
    model._ctx.call("mocha_column_stats", _ptr(cntf), C.c_int64(cntf.shape[0]), _ptr(mean), _ptr(std), _stream())
    return {"encoded": encoded, "cnt": cntf, "cnt_mean": mean, "cnt_std": std}


def save_bank(path: str, bank: dict, range_starts=None, range_stops=None, action_label=None, norm_path: Optional[str] = None):
    n = bank["encoded"].shape[0]
    np.savez_compressed(path, encoded=bank["encoded"].cpu().numpy(), cnt=bank["cnt"].cpu().numpy(),
                        range_starts=np.asarray([0] if range_starts is None else range_starts),
                        range_stops=np.asarray([n] if range_stops is None else range_stops),
                        action_label=np.asarray([] if action_label is None else action_label))
    if norm_path:
        np.savez_compressed(norm_path, mean=bank["cnt_mean"].cpu().numpy(), std=bank["cnt_std"].cpu().numpy())


def load_bank(path: str, norm_path: Optional[str] = None) -> dict:
    z = np.load(path, allow_pickle=True)
    out = {k: z[k] for k in ("encoded", "cnt", "range_starts", "range_stops", "action_label") if k in z}
    if norm_path:
        n = np.load(norm_path, allow_pickle=True)
        out["cnt_mean"], out["cnt_std"] = n["mean"], n["std"]
    return out


def reduce_matches(dist_local: torch.Tensor, idx_global: torch.Tensor):
    """All-gather per-rank (distance, global index) candidates and keep the nearest per query; ties go to the lowest
    index, as in the single-GPU matcher.  Works on any backend (tested with gloo)."""
    if not torch.distributed.is_initialized() or torch.distributed.get_world_size() == 1:
        return dist_local, idx_global
    world = torch.distributed.get_world_size()
    ds = [torch.empty_like(dist_local) for _ in range(world)]
    ix = [torch.empty_like(idx_global) for _ in range(world)]
    torch.distributed.all_gather(ds, dist_local.contiguous())
    torch.distributed.all_gather(ix, idx_global.contiguous())
    d = torch.stack(ds)                      # (world, Q)
    i = torch.stack(ix)
    # lexicographic (distance, index) minimum over ranks
    best_d = d.min(dim=0).values
    cand = torch.where(d == best_d[None], i, torch.full_like(i, torch.iinfo(i.dtype).max))
    return best_d, cand.min(dim=0).values


class ShardedContextBank:
    """Bank rows [lo, hi) of the z-scored cnt features live on this rank; ``encoded`` is replicated."""

    def __init__(self, model: Generator, cnt_nm_full_or_shard, encoded_full, n_total: int, bf16: bool = False):
        rank, _, world = D.env_rank()
        self.lo, self.hi = D.shard_bounds(n_total, world, rank)
        shard = cnt_nm_full_or_shard
        if shard.shape[0] == n_total:
            shard = shard[self.lo:self.hi]
        if shard.shape[0] != self.hi - self.lo:
            raise ValueError("cnt_nm must be the full bank or this rank's block")
        self.encoded = _dev_f32(encoded_full, model.device, (NTOK, DIM), "encoded")
        if self.encoded.shape[0] != n_total:
            raise ValueError("encoded must hold the full bank (it is replicated)")
        # the local ContextBank only needs `encoded` rows of its own block for its internal bookkeeping
        self.local = ContextBank(model, shard.contiguous(), self.encoded[self.lo:self.hi], bf16=bf16)
        self.model = model

    def query(self, query_nm):
        dist, idx = self.local.query(query_nm, k=1)
        return reduce_matches(dist[:, 0], idx[:, 0].to(torch.int64) + self.lo)

    def gather(self, idx):
        return self.encoded[idx.to(torch.int64)]
