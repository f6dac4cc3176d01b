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
    assert rec["value"] == max(r["frames_per_s"] for r in rec["sweep"]) == rec["sweep_best_frames_per_s"]      # full = 0: the best of the sweep
    assert rec["cores"] == rec["best"]["threads"]
    assert all(r["median_s"] > 0 and r["windows"] >= 8 for r in rec["sweep"])
    assert rec["single_thread_frames_per_s"] == rec["sweep"][0]["frames_per_s"]


def test_cpu_baseline_value_is_the_full_workload_run():
    """by default `value` is ONE run of the full workload at the sweep's best configuration (equal workloads on both
    sides of speedup_vs_cpu), the sample sweep stays in `sweep`."""
    import bench
    from mocha_sigasia2023_amd import synthetic, weights
    sd = weights.synthetic_state_dict(1777, 1.0, "mixamo")
    mean, std = synthetic.cnt_norm(7)
    rec = bench.cpu_baseline(sd, 22, 8, mean, std, full=12)
    full = rec["full_workload"]
    assert full["windows"] == 12 and rec["value"] == full["frames_per_s"] and full["threads"] == rec["cores"]
    assert rec["sample"].startswith("value: ONE run of the full workload, 12 src + 12 cha windows")
    assert rec["sweep_best_frames_per_s"] == max(r["frames_per_s"] for r in rec["sweep"])


def test_kernel_source_hash_ignores_comments_and_layout(tmp_path):
    """the hash that ties a committed PMC summary to the library covers what the GPU runs - comment and white-space
    edits do not change it, a code edit does; literals are not mistaken for comments."""
    import bench
    a = 'int f(int x) { // add one\n    return x + 1;   /* really */\n}\nconst char* s = "// not a comment /* x */"; char c = \'"\';\n'
    b = 'int f(int x) {\n  return x + 1;\n}   // moved\nconst char* s = "// not a comment /* x */";\nchar c = \'"\';'
    c = 'int f(int x) { return x + 2; }\nconst char* s = "// not a comment /* x */"; char c = \'"\';'
    assert bench.strip_c_comments(a) == bench.strip_c_comments(b) != bench.strip_c_comments(c)
    assert '"// not a comment /* x */"' in bench.strip_c_comments(a) and "add one" not in bench.strip_c_comments(a)
    h = bench.kernel_source_sha16()
    assert len(h) == 16 and h == bench.kernel_source_sha16()


def test_power_sampler_never_fails_the_measurement():
    """bench.py samples rocm-smi (read-only) during its sustained region; without a GPU / without rocm-smi the record is empty, not an error."""
    import time
    import bench
    with bench.PowerSampler(0.02) as ps:
        time.sleep(0.1)
    rec = ps.record()
    assert "samples" in rec and (rec["samples"] == 0 or rec["avg_w"] > 0)
