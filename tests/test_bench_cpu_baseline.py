"""bench.py's cpu_baseline leg (the reference's only timing code is model.py:311-318) on a tiny sample: a thread sweep
with warm-up and medians, the best configuration as `value`, every configuration in `sweep`, CPU model and core count in the record."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def test_cpu_baseline_record_is_a_sweep():
    import bench
    from mocha_sigasia2023_amd import synthetic, weights
    sd = weights.synthetic_state_dict(1777, 1.0, "mixamo")
    mean, std = synthetic.cnt_norm(7)
    before = torch.get_num_threads()
    rec = bench.cpu_baseline(sd, 22, 8, mean, std)
    assert torch.get_num_threads() == before                                   # the sweep restores the thread count
    assert rec["kind"] == "port" and rec["unit"] == "frames/s" and rec["value"] > 0
    assert rec["cpu_model"] and rec["os_cpu_count"] == os.cpu_count()
    threads = [r["threads"] for r in rec["sweep"]]
    assert threads[0] == 1 and before in threads and len(rec["sweep"]) >= 3    # one thread, all threads, and the whole-clip batch
    assert rec["value"] == max(r["frames_per_s"] for r in rec["sweep"])
    assert rec["cores"] == rec["best"]["threads"]
    assert all(r["median_s"] > 0 and r["windows"] >= 8 for r in rec["sweep"])
    assert rec["single_thread_frames_per_s"] == rec["sweep"][0]["frames_per_s"]
