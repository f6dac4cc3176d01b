"""Character feature bank: build, statistics, on-disk format, and a bank-sharded matcher.

* ``build_bank``   — what collect_CVAE_feature_action.py:167-180 / compute_cnt_norm.py:157-175 do: batch-encode all
  60-frame windows of a character database into ``encoded`` and ``cnt`` and take the per-(token, channel) mean / std
  of ``cnt`` over the entries.  All arithmetic runs in the HIP kernels; this module only loops and stores.
* ``save_bank`` / ``load_bank`` — the reference's ``np.savez_compressed(encoded=, cnt=, range_starts=, range_stops=,
  action_label=)`` feature file (collect_CVAE_feature_action.py:185-189) and ``cnt_norm.npz`` (mean, std;
  compute_cnt_norm.py:178-179).
* ``load_database`` / ``write_database`` — the reference's ``database.bin`` (etc/utils.py:144-190, written by
  preprocess/generate_database_bin.py:228-246); ``collect_windows`` — the bank scripts' step-1 windowing over the
  clips of one character (collect_CVAE_feature_action.py:104-133); ``build_bank_from_database`` chains them with the
  device featurisation and ``build_bank``: database.bin -> windows -> X -> encoded / cnt -> the reference's ``.npz`` files.
* ``BatchPipeline`` — consecutive batches overlapped: a few contexts of one process, each with its own stream, characterize
  alternate batches against the same bank.  A mid-size batch (the per-GPU share of BASELINE configs[3]: 128 windows) is a latency
  chain of ~37 launches that leaves most of the chip idle; independent batches fill each other's gaps (128 windows: 121 -> 160 k
  frames/s with three contexts, profiles/r05/f_mid_pipeline.txt).
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


def load_database(filename: str) -> dict:
    """The reference's motion database file (etc/utils.py:144-190): little-endian blocks, each with a uint32 header —
    bone positions, velocities (nframes, nbones, 3) f32, rotations (nframes, nbones, 4) f32 (w, x, y, z), angular
    velocities (nframes, nbones, 3) f32, bone parents (nbones) i32, range starts / stops (nranges) i32, style labels,
    action ("content") labels (nranges) i32, contact states (nframes, ncontacts) int8.  The reference returns the action
    labels under ``content_labels`` (:172-173) while its bank script reads ``action_labels``
    (collect_CVAE_feature_action.py:81); both keys are provided."""
    import struct
    with open(filename, "rb") as f:
        buf = f.read()
    off = 0

    def u32(n):
        nonlocal off
        v = struct.unpack_from("<" + "I" * n, buf, off)
        off += 4 * n
        return v

    def arr(dtype, shape):
        nonlocal off
        count = int(np.prod(shape))
        a = np.frombuffer(buf, dtype=dtype, count=count, offset=off).reshape(shape)
        off += count * np.dtype(dtype).itemsize
        return a

    out = {}
    for key, width in (("bone_positions", 3), ("bone_velocities", 3), ("bone_rotations", 4), ("bone_angular_velocities", 3)):
        nframes, nbones = u32(2)
        out[key] = arr("<f4", (nframes, nbones, width))
    out["bone_parents"] = arr("<i4", (u32(1)[0],))
    for key in ("range_starts", "range_stops", "style_labels", "content_labels"):
        out[key] = arr("<i4", (u32(1)[0],))
    out["action_labels"] = out["content_labels"]
    nframes, ncontacts = u32(2)
    out["contact_states"] = arr("i1", (nframes, ncontacts))
    if off != len(buf):
        raise ValueError(f"{filename}: {len(buf) - off} trailing bytes (not a database.bin of this layout)")
    return out


def write_database(filename: str, db: dict) -> None:
    """Inverse of ``load_database``: the block sequence of preprocess/generate_database_bin.py:228-246."""
    import struct
    f32 = lambda k: np.ascontiguousarray(db[k], dtype="<f4")
    i32 = lambda k: np.ascontiguousarray(db[k], dtype="<i4")
    pos = f32("bone_positions")
    nframes, nbones = pos.shape[:2]
    labels = "action_labels" if "action_labels" in db else "content_labels"
    contacts = np.ascontiguousarray(db["contact_states"]).astype(np.uint8)
    with open(filename, "wb") as f:
        for k in ("bone_positions", "bone_velocities", "bone_rotations", "bone_angular_velocities"):
            f.write(struct.pack("<II", nframes, nbones) + f32(k).tobytes())
        f.write(struct.pack("<I", nbones) + i32("bone_parents").tobytes())
        for k in ("range_starts", "range_stops", "style_labels", labels):
            f.write(struct.pack("<I", len(db["range_starts"])) + i32(k).tobytes())
        f.write(struct.pack("<II", nframes, contacts.shape[1]) + contacts.tobytes())


def collect_windows(db: dict, style_labels, action_labels, window: int = 60) -> dict:
    """Step-1 windows of every clip whose style and action are wanted (collect_CVAE_feature_action.py:104-133): a clip
    [start, stop) yields the windows [start + j - window, start + j) for j = window .. stop - start - 1, i.e.
    stop - start - window of them (the last possible window is not taken, as in the reference's ``range(window,
    total_frames)``).  Returns the windows' first frames (``starts``, global frame indices), the per-window action label
    and the per-clip [range_starts, range_stops) over the window list (the reference's cha_range_starts / _stops)."""
    styles, actions = set(int(s) for s in style_labels), set(int(a) for a in action_labels)
    starts, labels, rs, re_ = [], [], [], []
    for i in range(len(db["range_starts"])):
        if int(db["style_labels"][i]) not in styles or int(db["action_labels"][i]) not in actions:
            continue
        start, stop = int(db["range_starts"][i]), int(db["range_stops"][i])
        n = max(stop - start - window, 0)
        starts.extend(start + j - window for j in range(window, stop - start))
        labels.extend([int(db["action_labels"][i])] * n)
        off = re_[-1] if re_ else 0
        rs.append(off); re_.append(off + (stop - start - window))
    return {"starts": np.asarray(starts, dtype=np.int64), "action_label": np.asarray(labels, dtype=np.int32),
            "range_starts": np.asarray(rs, dtype=np.int32), "range_stops": np.asarray(re_, dtype=np.int32)}


def build_bank_from_database(model: Generator, database, style_labels, action_labels, window: int = 60, batch: int = 1024) -> dict:
    """database.bin (path or ``load_database`` dict) -> the character bank of collect_CVAE_feature_action.py:83-189 and
    compute_cnt_norm.py:157-179: windows of the wanted clips, featurised (FK, re-rooting, z-score with the pose norm of
    ``model.set_pose_norm``) and encoded on the device.  Returns ``build_bank``'s dict plus ``range_starts``,
    ``range_stops``, ``action_label`` and ``starts``; ``save_bank`` writes the reference's ``.npz`` files from it."""
    db = load_database(database) if isinstance(database, str) else database
    w = collect_windows(db, style_labels, action_labels, window)
    if not getattr(model, "_has_pose_norm", False):
        raise RuntimeError("build_bank_from_database needs the pose norm: call model.set_pose_norm(X_mean, X_std, Y_mean, Y_std) first")
    idx = w["starts"][:, None] + np.arange(window)[None]                       # (N, window) frame indices
    enc, cnt = [], []
    for s in range(0, len(idx), batch):
        ii = idx[s:s + batch]
        rot, pos, vel, ang = (torch.from_numpy(np.ascontiguousarray(db[k][ii], dtype=np.float32)) for k in
                              ("bone_rotations", "bone_positions", "bone_velocities", "bone_angular_velocities"))
        X_raw = model.featurize(rot, pos, vel, ang)
        e, c = model.encode(X_raw, raw=True)
        enc.append(e); cnt.append(c)
    if not enc:
        raise ValueError("no clip of the database matches the wanted style / action labels")
    encoded, cntf = torch.cat(enc), torch.cat(cnt)
    mean = torch.empty((NTOK, DIM), dtype=torch.float32, device=model.device)
    std = torch.empty_like(mean)
    model._ctx.call("mocha_column_stats", _ptr(cntf), C.c_int64(cntf.shape[0]), _ptr(mean), _ptr(std), _stream())
    return {"encoded": encoded, "cnt": cntf, "cnt_mean": mean, "cnt_std": std, **w}


def save_bank(path: str, bank: dict, range_starts=None, range_stops=None, action_label=None, norm_path: Optional[str] = None):
    n = bank["encoded"].shape[0]
    range_starts = bank.get("range_starts") if range_starts is None else range_starts
    range_stops = bank.get("range_stops") if range_stops is None else range_stops
    action_label = bank.get("action_label") if action_label is None else action_label
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


class BatchPipeline:
    """Overlap consecutive batches of ``ContextBank.characterize`` (test_fullframework.py:188-194, 438-443, 465-467 on many clips).

    ``contexts`` Generators with the same weights, each on its own HIP stream, take the batches in turn; the bank's rows are
    borrowed by every context (one copy of ``cnt_nm`` / ``encoded``; the derived data - centroid, norms, bf16 copy, decoder
    constants - per context).  The windows of a batch are independent of every other batch's, so nothing is exchanged.
    ``characterize`` returns as soon as the work is enqueued; ``join()`` makes the caller's stream wait for everything
    submitted so far (call it before reading results on the caller's stream, or synchronise the device)."""

    def __init__(self, state_dict, cha_cnt_nm, cha_encoded, layout: str = "mocha", device="cuda:0", contexts: int = 3,
                 bf16: bool = False, config=None, options=None):
        if contexts < 1:
            raise ValueError("contexts must be >= 1")
        self.device = torch.device(device)
        self.models = [Generator(config, layout=layout, device=self.device).load_state_dict(state_dict).eval() for _ in range(contexts)]
        for m in self.models:
            for k, v in (options or {}).items():
                m.set_option(k, v)
        self.banks = [ContextBank(m, cha_cnt_nm, cha_encoded, bf16=bf16) for m in self.models]
        self.streams = [torch.cuda.Stream(device=self.device) for _ in range(contexts)]
        self._next = 0

    def characterize(self, src_X, cnt_mean, cnt_std, return_index: bool = False, raw: bool = False):
        k = self._next % len(self.models)
        self._next += 1
        st = self.streams[k]
        st.wait_stream(torch.cuda.current_stream(self.device))          # the inputs were produced on the caller's stream
        for t in (src_X, cnt_mean, cnt_std):
            if isinstance(t, torch.Tensor) and t.is_cuda:
                t.record_stream(st)                                     # read on the side stream: the caller may drop them right after this call
        with torch.cuda.stream(st):
            out = self.banks[k].characterize(src_X, cnt_mean, cnt_std, return_index=return_index, raw=raw)
        for t in (out if isinstance(out, tuple) else (out,)):
            t.record_stream(torch.cuda.current_stream(self.device))     # allocated on the side stream, read on the caller's after join()
        return out

    def join(self):
        cur = torch.cuda.current_stream(self.device)
        for st in self.streams:
            cur.wait_stream(st)


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
        # the local ContextBank only needs `encoded` rows of its own block for its internal bookkeeping; its per-entry decoder constants
        # (+ 92 KB per entry at mocha_bank_set) would never be read - gather() indexes the replicated tensor and the decoder gets the copy
        self.local = ContextBank(model, shard.contiguous(), self.encoded[self.lo:self.hi], bf16=bf16, dec_cache=False)
        self.model = model

    def query(self, query_nm):
        dist, idx = self.local.query(query_nm, k=1)
        return reduce_matches(dist[:, 0], idx[:, 0].to(torch.int64) + self.lo)

    def gather(self, idx):
        return self.encoded[idx.to(torch.int64)]
