"""One rank of tests/test_multirank_standin.py: an ordinary process on GPU 0 whose context talks to its peers through the
test-only RCCL stand-in (MOCHA_RCCL_LIBRARY), with torch.distributed/gloo as the side channel for the unique id.
Usage: worker.py <mode> <out_dir>; RANK / WORLD_SIZE / MASTER_* in the environment."""
import ctypes as C
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)

from mocha_sigasia2023_amd import ContextBank, Generator, distributed as D, weights  # noqa: E402

DIM = 90 * 256


def bank_data(seed, N):
    r = np.random.Generator(np.random.PCG64(seed))
    nm = r.standard_normal((N, DIM)).astype(np.float32)
    enc = r.standard_normal((N, 90, 256)).astype(np.float32)
    return nm, enc


def queries(seed, Q, nm):
    r = np.random.Generator(np.random.PCG64(seed))
    rows = r.integers(0, nm.shape[0], Q)
    return (nm[rows] + 0.05 * r.standard_normal((Q, DIM))).astype(np.float32)


def view(model, bf16):
    """The context's current bank and what the library derived from it, copied to the host (mocha_bank_view)."""
    ptrs = [C.c_void_p() for _ in range(5)]
    n = C.c_int64()
    model._ctx.call("mocha_bank_view", *[C.byref(p) for p in ptrs], C.byref(n))
    torch.cuda.synchronize()
    hip = C.CDLL("libamdhip64.so.7")
    N = n.value

    def pull(p, shape, dt):
        a = np.empty(shape, dt)
        assert hip.hipMemcpy(a.ctypes.data_as(C.c_void_p), p, a.nbytes, 2) == 0
        return a
    out = {"N": N, "cnt": pull(ptrs[0], (N, DIM), np.float32), "enc": pull(ptrs[1], (N, DIM), np.float32),
           "centroid": pull(ptrs[2], (DIM,), np.float32), "norm": pull(ptrs[3], (N,), np.float32)}
    if bf16:
        assert ptrs[4].value
        out["bf16"] = pull(ptrs[4], (N, DIM), np.uint16)
    else:
        assert not ptrs[4].value
    return out


def run_broadcast(model, rank, world, out_dir, tag, seed, N, root, bf16):
    nm, enc = bank_data(seed, N)
    bank = ContextBank(model, torch.from_numpy(nm), torch.from_numpy(enc), bf16=bf16) if rank == root else None
    got = D.bank_broadcast(model, bank, N, root=root, bf16=bf16)
    torch.cuda.synchronize()
    res = view(model, bf16)
    for name, Q in (("few", 5), ("many", 40)):                      # streaming scan and many-query path
        q = torch.from_numpy(queries(seed + 1, Q, nm))
        d, i = got.query(q)
        res[f"idx_{name}"] = i[:, 0].cpu().numpy()
        res[f"dist_{name}"] = d[:, 0].cpu().numpy()
    res["gather"] = got.gather(torch.tensor([0, N - 1], dtype=torch.int32)).cpu().numpy()
    np.savez(os.path.join(out_dir, f"{tag}_rank{rank}.npz"), **res)


def main():
    mode, out_dir = sys.argv[1], sys.argv[2]
    rank, _, world = D.env_rank()
    D.init("gloo")
    model = Generator(device="cuda:0").load_state_dict(weights.synthetic_state_dict(11, 1.0)).eval()
    if mode == "broadcast":
        # N = 5: with 7 ranks 5 * 23040 is not a multiple of the world (23040 = 2^9 3^2 5 divides by 2, 3, 4, 5, 6, 8), so the
        # tail broadcast runs; the second, larger bank re-allocates the receivers' buffers while a received bank is current;
        # the third moves the root
        run_broadcast(model, rank, world, out_dir, "a", 100, 5, 0, False)
        run_broadcast(model, rank, world, out_dir, "b", 200, 67, 0, True)
        run_broadcast(model, rank, world, out_dir, "c", 300, 33, world - 1, False)
    elif mode == "mismatch":
        nm, enc = bank_data(1, 9)
        bank = ContextBank(model, torch.from_numpy(nm), torch.from_numpy(enc), bf16=False) if rank == 0 else None
        msgs = []
        # flags disagree with the root's bank; the entry count does; ONE non-root rank alone disagrees on the count
        for kw in ({"bf16": True}, {"n": 10}, {"n": 10 if rank == world - 1 else 9}):
            try:
                D.bank_broadcast(model, bank, kw.get("n", 9), root=0, bf16=kw.get("bf16", False))
                msgs.append("no error")
            except RuntimeError as e:
                msgs.append(str(e))
        got = D.bank_broadcast(model, bank, 9, root=0, bf16=False)   # and the communicator is still usable afterwards
        idx = got.query(torch.from_numpy(nm), return_distance=False)[:, 0].cpu().tolist()
        with open(os.path.join(out_dir, f"mismatch_rank{rank}.txt"), "w") as f:
            f.write("\n".join(msgs + [str(idx)]))
    elif mode == "fail_send":
        nm, enc = bank_data(2, 12)
        bank = ContextBank(model, torch.from_numpy(nm), torch.from_numpy(enc)) if rank == 0 else None
        try:
            D.bank_broadcast(model, bank, 12, root=0)
            msg = "no error"
        except RuntimeError as e:
            msg = str(e)
        got = D.bank_broadcast(model, bank, 12, root=0)              # the failed scatter left no open group behind
        idx = got.query(torch.from_numpy(nm), return_distance=False)[:, 0].cpu().tolist()
        with open(os.path.join(out_dir, f"fail_send_rank{rank}.txt"), "w") as f:
            f.write(msg + "\n" + str(idx))
    else:
        raise SystemExit(f"unknown mode {mode}")
    D.barrier()
    torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
