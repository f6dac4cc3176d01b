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
    if "MASTER_PORT" not in os.environ:
        if world > 1:
            raise RuntimeError("MASTER_PORT is not set: launch the ranks with torch.distributed.run (or bench.py --gpus N), "
                               "which choose a port per job")
        os.environ["MASTER_PORT"] = str(20000 + os.getpid() % 20000)      # single rank: any free-ish port, per process
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


def _host_staged() -> bool:
    """True when the default group cannot move device tensors (gloo): collectives then go through host copies."""
    return dist.is_initialized() and dist.get_backend() == "gloo"


def broadcast_(tensors: Iterable[torch.Tensor], src: int = 0) -> None:
    """In-place broadcast of the bank tensors (cnt_nm, encoded, cnt norm, or the raw character clip)."""
    if dist.is_initialized():
        for t in tensors:
            if _host_staged() and t.is_cuda:
                h = t.cpu()
                dist.broadcast(h, src=src)
                t.copy_(h)
            else:
                dist.broadcast(t, src=src)


def init_comm(model) -> None:
    """Give the model's context its RCCL communicator (``mocha_comm_init``): rank 0 draws the 128-byte unique id through
    the C ABI and the default process group (any backend) ships it to the other ranks."""
    import ctypes as C
    rank, _, world = env_rank()
    if getattr(model, "_comm_ready", False):
        return
    # one RCCL per process: PyTorch bundles its own copy (the one torch.distributed's "nccl" backend drives); use it if present.
    # MOCHA_RCCL_LIBRARY names another file (the tests load a shared-memory stand-in to run several ranks on one GPU).
    override = os.environ.get("MOCHA_RCCL_LIBRARY")
    bundled = override or os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
    if override and not os.path.exists(override):
        raise RuntimeError(f"MOCHA_RCCL_LIBRARY={override}: no such file")
    if os.path.exists(bundled):
        model._ctx.lib.mocha_set_rccl_library(bundled.encode())        # status -3 = already resolved: keep what is loaded
    buf = (C.c_char * 128)()
    if rank == 0:
        model._ctx.call("mocha_comm_unique_id", C.cast(buf, C.c_void_p))
    if world > 1:
        obj = [bytes(buf.raw)]
        dist.broadcast_object_list(obj, src=0)
        buf = (C.c_char * 128).from_buffer_copy(obj[0])
    model._ctx.call("mocha_comm_init", C.cast(buf, C.c_void_p), world, rank)
    model._comm_ready = True


def comm_info(model) -> dict:
    """``mocha_comm_info``: what the context's communicator really is - ranks as RCCL reports them (ncclCommCount), RCCL's version,
    the file its entry points were resolved from, the context's device and PCI bus id."""
    import ctypes as C
    from ._C import mocha_comm_info_t
    info = mocha_comm_info_t()
    model._ctx.call("mocha_comm_info", C.byref(info))
    v = int(info.rccl_version)
    return {"nranks": int(info.nranks), "rank": int(info.rank), "rccl_version_code": v,
            "rccl_version": f"{v // 10000}.{v // 100 % 100}.{v % 100}", "library": info.library.decode(),
            "device": int(info.device), "pci_bus_id": info.pci_bus_id.decode()}


def bank_broadcast(model, bank, n_entries: int, root: int = 0, bf16: bool = False):
    """``mocha_bank_broadcast``: the root's ContextBank becomes every rank's current bank over RCCL / xGMI (scatter +
    all-gather of cnt_nm and encoded; derived data recomputed locally).  Returns the rank's bank handle."""
    import ctypes as C
    from .generator import ContextBank, _stream
    rank, _, _ = env_rank()
    init_comm(model)
    if rank == root:
        bank.activate()
    model._ctx.call("mocha_bank_broadcast", C.c_void_p(0), root, C.c_int64(n_entries), 2 if bf16 else 0, _stream())
    return bank if rank == root else ContextBank.received(model, n_entries, bf16)


def all_gather_rows(local: torch.Tensor, n_total: int) -> torch.Tensor:
    """Concatenate per-rank row blocks produced with ``shard_bounds`` back into (n_total, ...)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return local
    world = dist.get_world_size()
    biggest = -(-n_total // world)
    cdev = torch.device("cpu") if _host_staged() else local.device
    pad = torch.zeros((biggest,) + tuple(local.shape[1:]), dtype=local.dtype, device=cdev)
    pad[: local.shape[0]] = local
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad)
    out = []
    for r, p in enumerate(parts):
        lo, hi = shard_bounds(n_total, world, r)
        out.append(p[: hi - lo])
    return torch.cat(out).to(local.device)


def max_over_ranks(value: float, device: torch.device) -> float:
    if not dist.is_initialized():
        return value
    t = torch.tensor([value], dtype=torch.float64, device=torch.device("cpu") if _host_staged() else device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def barrier() -> None:
    if dist.is_initialized():
        dist.barrier()
