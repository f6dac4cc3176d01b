"""The multi-rank C-ABI path with MORE THAN ONE RANK, on a one-GPU box (run with -m gpu).

mocha_bank_broadcast's scatter + all-gather + tail broadcast, the receiving side's allocation branch, ContextBank.received
and bench.py's N > 1 branches need several ranks; real RCCL refuses two ranks on one device.  The library resolves RCCL
through a function table (mocha_set_rccl_library), so these tests load tests/rccl_standin/librccl_standin.so - the thirteen
entry points over POSIX shared memory + hipMemcpy, test infrastructure only - and start every rank as a fresh process on
GPU 0.  Split being tested: test_fullframework.py:148-158, 440-443, 465-467 (windows are independent; the bank is read-only).
"""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STANDIN_DIR = os.path.join(REPO, "tests", "rccl_standin")
STANDIN = os.path.join(STANDIN_DIR, "librccl_standin.so")
WORKER = os.path.join(STANDIN_DIR, "worker.py")
DIM = 90 * 256


@pytest.fixture(scope="module")
def standin():
    src = os.path.join(STANDIN_DIR, "rccl_standin.cpp")
    if not os.path.exists(STANDIN) or os.path.getmtime(STANDIN) < os.path.getmtime(src):
        subprocess.run(["make", "-C", STANDIN_DIR], check=True)
    return STANDIN


def _port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _env(standin, **kw):
    env = dict(os.environ, MOCHA_RCCL_LIBRARY=standin, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MOCHA_FORCE_DIST"):
        env.pop(k, None)
    env.update(kw)
    return env


def run_ranks(standin, world, mode, out_dir, extra_env=None, timeout=600):
    port = _port()
    procs = []
    for r in range(world):
        env = _env(standin, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   **(extra_env or {}))
        procs.append(subprocess.Popen([sys.executable, WORKER, mode, str(out_dir)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = []
    try:
        for p in procs:
            outs.append(p.communicate(timeout=timeout)[0])
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    for r, p in enumerate(procs):
        assert p.returncode == 0, f"rank {r} of {world} ({mode}) exited {p.returncode}:\n{outs[r][-3000:]}"


def _bank(seed, N):
    r = np.random.Generator(np.random.PCG64(seed))
    return r.standard_normal((N, DIM)).astype(np.float32), r.standard_normal((N, 90, 256)).astype(np.float32)


@pytest.mark.timeout(1200)
@pytest.mark.parametrize("world", [2, 4, 7, 8])
def test_bank_broadcast_reaches_every_rank_bit_for_bit(standin, tmp_path, world):
    """Every rank's bank - rows, encoded entries, centroid, centred row norms, the centred bf16 copy - equals the root's bit
    for bit; query() on every rank returns the root's indices and distances (scan and many-query path); the receivers'
    buffers survive a growth and a change of root.  world = 4 and 8 are the rank counts of the driver's 1 -> 8 curve (VERDICT r5
    item 2); world = 7 is the one world size below 9 that does not divide N * 23040, so the tail broadcast runs there."""
    run_ranks(standin, world, "broadcast", tmp_path)
    for tag, seed, N, root, bf16 in (("a", 100, 5, 0, False), ("b", 200, 67, 0, True), ("c", 300, 33, world - 1, False)):
        nm, enc = _bank(seed, N)
        z = [np.load(tmp_path / f"{tag}_rank{r}.npz") for r in range(world)]
        ref = z[root]
        assert np.array_equal(ref["cnt"], nm) and np.array_equal(ref["enc"].reshape(N, 90, 256), enc)       # the root's bank is the data
        for r in range(world):
            assert int(z[r]["N"]) == N
            for k in ref.files:
                assert np.array_equal(z[r][k], ref[k]), f"bank {tag}: rank {r} differs from the root in '{k}'"
        # and the answers are right, not merely equal: float64 search over the rows (bf16: agreement is checked by the parity tests)
        if not bf16:
            for name, Q in (("few", 5), ("many", 40)):
                r2 = np.random.Generator(np.random.PCG64(seed + 1))          # worker.queries(seed + 1, Q, nm)
                rows = r2.integers(0, N, Q)
                q = (nm[rows] + 0.05 * r2.standard_normal((Q, DIM))).astype(np.float32)
                d2 = ((q[:, None, :].astype(np.float64) - nm[None].astype(np.float64)) ** 2).sum(-1)
                assert np.array_equal(ref[f"idx_{name}"], d2.argmin(1))
        assert np.array_equal(ref["gather"], enc[[0, N - 1]])
    if world == 7:
        from mocha_sigasia2023_amd import _C
        import ctypes as C
        out = (C.c_int64 * 4)()
        _C.load_library().mocha_bcast_plan(5 * DIM, 7, 0, out)
        assert out[3] > 0, "the world-7 case is meant to exercise the tail broadcast"


@pytest.mark.timeout(600)
def test_disagreeing_ranks_fail_loudly_on_every_rank(standin, tmp_path):
    """A bf16 / fp32 flag mismatch, a wrong entry count on every rank, and a wrong entry count on ONE non-root rank are caught by
    the all-gathered headers every rank checks: all ranks raise, none hangs in a collective the others never entered, and the
    communicator still works afterwards."""
    run_ranks(standin, 3, "mismatch", tmp_path)
    for r in range(3):
        flags_msg, count_msg, lone_msg, idx = open(tmp_path / f"mismatch_rank{r}.txt").read().split("\n")
        assert "MOCHA_BANK_BF16" in flags_msg and "opposite" in flags_msg
        assert "no current bank" in count_msg
        # one rank alone disagrees: the all-gathered headers fail EVERY rank - the others name it, it sees the root's announcement
        assert ("rank 2 asks for 10 entries" in lone_msg) if r != 2 else ("it announced 9" in lone_msg), lone_msg
        assert idx == str(list(range(9)))


@pytest.mark.timeout(600)
def test_failed_send_closes_the_group(standin, tmp_path):
    """an ncclSend that fails inside the scatter group must not leave the RCCL group open.  The stand-in
    fails rank 0's first ncclSend once; the call fails on every rank and the next broadcast succeeds."""
    run_ranks(standin, 3, "fail_send", tmp_path, extra_env={"MOCHA_STANDIN_FAIL_SEND": "0"})
    for r in range(3):
        msg, idx = open(tmp_path / f"fail_send_rank{r}.txt").read().split("\n")
        assert msg != "no error" and ("ncclSend" in msg or "unmatched" in msg or "bank scatter" in msg), msg
        assert idx == str(list(range(12)))


def _bench(standin, args, **env):
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py")] + args, env=_env(standin, **env), capture_output=True, text=True, timeout=1500)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    return json.loads(lines[0])


@pytest.fixture(scope="module")
def bank4k_one_rank(standin):
    """ONE 1-rank run of `bench.py --workload bank4k` shared by the tests that compare an N-rank split with it."""
    return _bench(standin, ["--workload", "bank4k", "--steps", "1", "--warmup", "1"])


@pytest.mark.timeout(3000)
def test_bench_bank4k_split_two_ways_reproduces_the_one_rank_indices(standin, bank4k_one_rank):
    """BASELINE configs[3] in small: bench.py --workload bank4k with the 1024 windows split over 2 ranks through shard_bounds
    and the 4096-entry bf16 bank broadcast through the C ABI gives exactly the indices of the 1-rank run.  The 2-rank run passes
    --options (ranks other than 0 used to raise on the options record before the rccl collective) - an option at its default
    value, so the arithmetic is the 1-rank run's."""
    one = bank4k_one_rank
    two = _bench(standin, ["--workload", "bank4k", "--gpus", "2", "--steps", "1", "--warmup", "1", "--options", "gemm_persistent=768"],
                 MOCHA_BENCH_ONE_GPU="1", MOCHA_BENCH_BACKEND="gloo")
    assert one["n_gpus"] == 1 and two["n_gpus"] == 2
    assert two["config"]["options"] == {"gemm_persistent": 768} and two["rccl"]["nranks"] == 2
    assert two["bank_broadcast_ms"] is not None and two["bank_broadcast_ms"] > 0
    assert two["idx_crc32"] == one["idx_crc32"] and two["idx_head"] == one["idx_head"]
    assert abs(two["max_abs_Y"] - one["max_abs_Y"]) < 1e-4
    # per-window fingerprints of 16 windows spread over both shards: the right window in the right place (the two runs batch
    # 1024 and 512 windows, so GEMM tile choices and with them the last bits may differ)
    assert len(one["y_fingerprint"]) == 16 and np.allclose(two["y_fingerprint"], one["y_fingerprint"], rtol=1e-5)
    assert len(set(np.round(one["y_fingerprint"], 3))) > 8                      # the windows really differ from each other


@pytest.mark.timeout(3000)
def test_bench_demo_two_ranks_runs_the_broadcast_and_probe(standin):
    """bench.py's default workload with N = 2 (bench.py: clip broadcast, C-ABI bank broadcast, probe of the received bank,
    max-over-ranks timing, per-rank rates)."""
    rec = _bench(standin, ["--gpus", "2", "--windows", "96", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-extras"],
                 MOCHA_BENCH_ONE_GPU="1", MOCHA_BENCH_BACKEND="gloo")
    assert rec["n_gpus"] == 2 and rec["scaling"] == "weak"
    assert rec.get("bank_broadcast_error") is None and rec["bank_broadcast_ms"] > 0
    assert len(rec["per_rank_frames_per_s"]) == 2 and rec["value"] > 0
    _check_rccl_record(rec["rccl"], 2, standin)
    _check_bank4k_subrecord(rec["bank4k"], 2)


def _check_rccl_record(r, world, standin):
    """The `rccl` record says what the communicator REALLY was: ranks from ncclCommCount, the library file, every rank's device -
    here the stand-in on one GPU, and the line says so (is_test_standin, test_hooks_in_env, distinct_devices == 1)."""
    assert r["nranks"] == world and r["nranks_agree"] is True
    assert os.path.realpath(r["library"]) == os.path.realpath(standin)
    assert r["is_test_standin"] is True and r["rccl_version_code"] == 1
    assert [p["rank"] for p in r["per_rank"]] == list(range(world))
    assert all(p["device"] == 0 and p["pci_bus_id"] for p in r["per_rank"]) and r["distinct_devices"] == 1
    assert r["test_hooks_in_env"].get("MOCHA_RCCL_LIBRARY") == standin and r["test_hooks_in_env"].get("MOCHA_BENCH_ONE_GPU") == "1"
    assert r["torch_side_channel_backend"] == "gloo"


def _check_bank4k_subrecord(b, world):
    """The configs[3] sub-record of the default N > 1 line: 1024 windows split by shard_bounds, the bank through the C ABI with
    its time and rate, the index checksum beside the known 1-GPU one."""
    assert b["n_gpus"] == world and b["scaling"] == "strong"
    assert sum(b["config"]["windows_per_gpu"]) == 1024 and len(b["config"]["windows_per_gpu"]) == world
    assert max(b["config"]["windows_per_gpu"]) - min(b["config"]["windows_per_gpu"]) <= 1
    bc = b["bank_broadcast"]
    assert bc["bytes_per_rank"] == 2 * 4096 * 23040 * 4 and bc["ms"] > 0 and bc["ms_first_call"] > 0
    assert abs(bc["GB/s_per_rank"] - bc["bytes_per_rank"] / bc["ms"] / 1e6) < 1e-6 * bc["GB/s_per_rank"]
    assert bc["xgmi_inbound_peak_GB/s"] == (world - 1) * 153.0 and 0 < bc["frac_of_xgmi_inbound_peak"]
    assert len(b["per_rank_frames_per_s"]) == world and b["value"] > 0
    assert "idx_crc32" in b and "idx_crc32_n1_known" in b
    if b["idx_crc32_n1_known"] is not None:
        assert b["idx_matches_n1"] is True, (b["idx_crc32"], b["idx_crc32_n1_known"])


@pytest.mark.timeout(3000)
def test_bench_default_line_eight_ranks_is_configs3(standin, bank4k_one_rank):
    """`python bench.py --gpus 8` - the last point of the driver's 1 -> 8 curve - through the stand-in: the weak-scaled headline over eight
    ranks AND the configs[3] sub-record exactly as BASELINE words it (1024 windows = 128 per GPU, the 4096-entry bank broadcast from rank 0),
    with the index checksum of the 1-rank run of the same build (the split: test_fullframework.py:148-158, 440-443)."""
    rec = _bench(standin, ["--gpus", "8", "--windows", "64", "--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--no-extras"],
                 MOCHA_BENCH_ONE_GPU="1", MOCHA_BENCH_BACKEND="gloo")
    assert rec["n_gpus"] == 8 and rec["bank_broadcast_error"] is None and rec["scaling"] == "weak"
    assert len(rec["per_rank_frames_per_s"]) == 8 and all(v > 0 for v in rec["per_rank_frames_per_s"])
    _check_rccl_record(rec["rccl"], 8, standin)
    assert rec["rccl"]["nranks"] == 8
    _check_bank4k_subrecord(rec["bank4k"], 8)
    assert rec["bank4k"]["config"]["windows_per_gpu"] == [128] * 8 and len(rec["bank4k"]["per_rank_frames_per_s"]) == 8
    one = bank4k_one_rank
    assert rec["bank4k"]["idx_crc32"] == one["idx_crc32"] and rec["bank4k"]["idx_head"] == one["idx_head"]
    assert np.allclose(rec["bank4k"]["y_fingerprint"], one["y_fingerprint"], rtol=1e-5)


@pytest.mark.timeout(3000)
def test_bench_bank4k_four_ranks_uneven_free_split(standin, bank4k_one_rank):
    """`bench.py --workload bank4k --gpus 4` (256 windows per GPU): the second interior point of the driver's curve, same checksum."""
    rec = _bench(standin, ["--workload", "bank4k", "--gpus", "4", "--steps", "1", "--warmup", "1"], MOCHA_BENCH_ONE_GPU="1", MOCHA_BENCH_BACKEND="gloo")
    assert rec["n_gpus"] == 4 and rec["config"]["windows_per_gpu"] == [256] * 4 and rec["rccl"]["nranks"] == 4
    assert rec["idx_crc32"] == bank4k_one_rank["idx_crc32"] and len(rec["per_rank_frames_per_s"]) == 4


@pytest.mark.timeout(600)
@pytest.mark.parametrize("workload", ["demo", "bank4k"])
def test_a_dying_rank_ends_the_job_within_seconds(standin, workload):
    """rank 1 dies right after the rendezvous (MOCHA_BENCH_FAIL_RANK) while rank 0 goes on into its first
    collective.  The launcher must exit non-zero promptly - not when the collective times out (gloo: 30 minutes) - name the
    rank and leave no rank behind."""
    import time
    t0 = time.monotonic()
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--windows", "32", "--steps", "1", "--warmup", "1",
                          "--no-cpu-baseline", "--no-extras", "--workload", workload],
                         env=_env(standin, MOCHA_BENCH_ONE_GPU="1", MOCHA_BENCH_BACKEND="gloo", MOCHA_BENCH_FAIL_RANK="1"),
                         capture_output=True, text=True, timeout=500)
    dt = time.monotonic() - t0
    assert out.returncode == 7, (out.returncode, out.stderr[-2000:])
    assert "rank 1 exited with status 7" in out.stderr and "injected failure" in out.stderr
    assert not [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert dt < 240, f"the launcher took {dt:.0f} s to give up"                    # start-up (imports, rendezvous) dominates; no 30-minute wait


@pytest.mark.timeout(1500)
def test_bench_under_torch_distributed_run_two_ranks(standin):
    """The driver's own launch line for N > 1 - `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
    --master-port P bench.py --gpus N ...` - with two ranks on GPU 0: bench.py reads RANK / LOCAL_RANK / WORLD_SIZE from the
    launcher's environment, prints ONE line on rank 0, whole-job value = sum over the ranks."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_port()), os.path.join(REPO, "bench.py"), "--gpus", "2", "--windows", "64", "--steps", "2", "--warmup", "1"]
    out = subprocess.run(cmd, env=_env(standin, MOCHA_BENCH_ONE_GPU="1", MOCHA_BENCH_BACKEND="gloo"), capture_output=True, text=True, timeout=1200)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["bank_broadcast_error"] is None and len(rec["per_rank_frames_per_s"]) == 2
    assert abs(rec["value"] - sum(rec["per_rank_frames_per_s"])) < 0.2 * rec["value"]
