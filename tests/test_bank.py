"""Row N4: bank build statistics / file format (GPU) and the bank-sharded match reduction (gloo, CPU)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from mocha_sigasia2023_amd import synthetic, weights


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from mocha_sigasia2023_amd import distributed as D
    from mocha_sigasia2023_amd.bank import reduce_matches
    from oracle import mocha_oracle as O
    D.init("gloo")
    N, Q = 37, 6
    bank = synthetic.token_features(1, N).reshape(N, -1)
    q = synthetic.token_features(2, Q).reshape(Q, -1)
    q[0] = bank[N - 1]; q[1] = bank[0]                       # exact hits in the last / first shard
    lo, hi = D.shard_bounds(N, world, rank)
    idx, dist = O.match_bruteforce(q, bank[lo:hi])           # the local scan (oracle stands in for the HIP matcher)
    d, i = reduce_matches(torch.from_numpy(dist.astype(np.float32)), torch.from_numpy(idx + lo))
    if rank == 0:
        np.savez(os.path.join(out_dir, "m.npz"), d=d.numpy(), i=i.numpy())
    D.barrier(); torch.distributed.destroy_process_group()


@pytest.mark.timeout(300)
def test_sharded_match_reduction_two_ranks(tmp_path):
    from oracle import mocha_oracle as O
    mp.spawn(_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    z = np.load(tmp_path / "m.npz")
    bank = synthetic.token_features(1, 37).reshape(37, -1)
    q = synthetic.token_features(2, 6).reshape(6, -1)
    q[0] = bank[36]; q[1] = bank[0]
    idx, dist = O.match_bruteforce(q, bank)
    assert np.array_equal(z["i"], idx)
    assert np.allclose(z["d"], dist, rtol=1e-6)


@pytest.mark.gpu
def test_build_bank_and_roundtrip(tmp_path):
    from mocha_sigasia2023_amd import ContextBank, Generator
    from mocha_sigasia2023_amd.bank import build_bank, load_bank, save_bank
    from oracle import mocha_oracle as O
    sd = weights.synthetic_state_dict(3, 1.0)
    model = Generator(device="cuda:0").load_state_dict(sd).eval()
    X = torch.from_numpy(synthetic.pose_windows(9, 21))
    bank = build_bank(model, X, batch=8)
    with torch.no_grad():
        enc, cnt = O.encode(O.to_torch_state(sd), X)
    assert float((bank["encoded"].cpu() - enc).abs().max()) < 1e-4 * float(enc.abs().max())
    assert np.allclose(bank["cnt_mean"].cpu().numpy(), cnt.numpy().mean(0), atol=2e-5)
    assert np.allclose(bank["cnt_std"].cpu().numpy(), cnt.numpy().std(0), atol=2e-5)       # compute_cnt_norm.py:174-175
    save_bank(str(tmp_path / "feat.npz"), bank, norm_path=str(tmp_path / "cnt_norm.npz"))
    z = load_bank(str(tmp_path / "feat.npz"), str(tmp_path / "cnt_norm.npz"))
    assert set(z) >= {"encoded", "cnt", "range_starts", "range_stops", "action_label", "cnt_mean", "cnt_std"}
    assert np.array_equal(z["encoded"], bank["encoded"].cpu().numpy())
    # the stored bank drives the matcher exactly like the freshly built one
    std = np.maximum(z["cnt_std"], 1e-3)
    nm = (z["cnt"] - z["cnt_mean"][None]) / std[None]
    idx = ContextBank(model, torch.from_numpy(nm).cuda(), torch.from_numpy(z["encoded"]).cuda()).query(torch.from_numpy(nm[:5]).cuda(), return_distance=False)
    assert idx[:, 0].cpu().tolist() == [0, 1, 2, 3, 4]


@pytest.mark.gpu
def test_batch_pipeline_equals_serial_calls():
    """BatchPipeline: three contexts on their own streams take alternate batches against one borrowed bank; every batch's poses and
    indices are those of the serial call, bit for bit (the same kernels on the same inputs; nothing is shared but read-only rows)."""
    from mocha_sigasia2023_amd import BatchPipeline, ContextBank, Generator
    dev = torch.device("cuda:0")
    sd = weights.synthetic_state_dict(5, 1.0)
    model = Generator(device=dev).load_state_dict(sd).eval()
    mean, std = (torch.from_numpy(a).to(dev) for a in synthetic.cnt_norm(7))
    enc, _, nm = model.encode(torch.from_numpy(synthetic.pose_windows(3, 70)).to(dev), mean, std)
    serial = ContextBank(model, nm, enc)
    pipe = BatchPipeline(sd, nm, enc, device=dev, contexts=3)
    batches = [torch.from_numpy(synthetic.pose_windows(20 + i, 5 + 9 * i)).to(dev) for i in range(7)]
    outs = [pipe.characterize(x, mean, std, return_index=True) for x in batches]
    pipe.join()
    torch.cuda.synchronize()
    for x, (Y, idx) in zip(batches, outs):
        Ys, ids = serial.characterize(x, mean, std, return_index=True)
        assert torch.equal(idx, ids) and torch.equal(Y, Ys)
    # a bank that lives in the context (as after mocha_bank_broadcast on a non-root rank) hands its rows on through mocha_bank_export
    owned = ContextBank(model, nm, enc, copy=True)
    got = ContextBank.received(model, owned.N)
    nm2, enc2 = got.tensors()
    torch.cuda.synchronize()
    assert torch.equal(nm2, nm.reshape(nm2.shape)) and torch.equal(enc2, enc)
    pipe2 = BatchPipeline(sd, nm2, enc2, device=dev, contexts=2)
    Y2, i2 = pipe2.characterize(batches[3], mean, std, return_index=True)
    pipe2.join(); torch.cuda.synchronize()
    assert torch.equal(Y2, outs[3][0]) and torch.equal(i2, outs[3][1])
