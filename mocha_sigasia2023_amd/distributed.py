"""Multi-GPU plumbing for the MOCHA path: one process per GPU, torch.distributed (backend
"nccl" is RCCL on ROCm; "gloo" on CPU for tests).

The path shards by independent units: every 60-frame source window is featurised, encoded,
matched against a read-only bank and decoded independently (test_fullframework.py:148-158,
440-443, 465-467).  So the only exchange is a one-time broadcast of the
character bank (or of the character clip it is built from) from the rank that owns it; there
is no collective inside the per-window loop.
"""
from __future__ import annotations

import os
from typing import Iterable, Tuple

import torch
import torch.distributed as dist


def env_rank() -> Tuple[int, int, int]:
    """(rank, local_rank, world_size) from the torchrun environment (1-process defaults)."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def init(backend: str, device: torch.device | None = None) -> Tuple[int, int]:
    """Initialise the default process group (rendezvous over 127.0.0.1 unless MASTER_ADDR is set)."""
    rank, _, world = env_rank()
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    if not dist.is_initialized():
        kw = {}
        if backend == "nccl" and device is not None:
            kw["device_id"] = device
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, world


def shard_bounds(n_items: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous block [lo, hi) of ``n_items`` units owned by ``rank`` (blocks differ by at most one)."""
    base, extra = divmod(n_items, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def broadcast_(tensors: Iterable[torch.Tensor], src: int = 0) -> None:
    """In-place broadcast of the bank tensors (cnt_nm, encoded, cnt norm, or the raw character clip)."""
    if dist.is_initialized():
        for t in tensors:
            dist.broadcast(t, src=src)


def all_gather_rows(local: torch.Tensor, n_total: int) -> torch.Tensor:
    """Concatenate per-rank row blocks produced with ``shard_bounds`` back into (n_total, ...)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return local
    world = dist.get_world_size()
    biggest = -(-n_total // world)
    pad = torch.zeros((biggest,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad)
    out = []
    for r, p in enumerate(parts):
        lo, hi = shard_bounds(n_total, world, r)
        out.append(p[: hi - lo])
    return torch.cat(out)


def max_over_ranks(value: float, device: torch.device) -> float:
    if not dist.is_initialized():
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def barrier() -> None:
    if dist.is_initialized():
        dist.barrier()
