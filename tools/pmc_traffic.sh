#!/bin/bash
# HBM-side traffic of every kernel of the default bench step: FETCH_SIZE and WRITE_SIZE in separate
# PMC passes (they do not fit one pass), kernel-trace only.  On gfx950 FETCH_SIZE counts 128-B
# requests at 64 B, so read bytes = 2 * FETCH_SIZE * 1024 for wide coalesced reads
# (MI355X_MICROARCH.md §HBM); WRITE_SIZE * 1024 is exact for 16-B-per-lane stores.
tag=${1:-r01}
cd /tmp; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/pmc_traffic_$tag; rm -rf $out; mkdir -p $out
for ctr in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $out/$ctr -o t -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --sustained-s 0 > $out/$ctr.stdout 2> $out/$ctr.stderr
done
python3 - $out <<'PY'
import csv, sys, glob, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: {"n": 0, "FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0})
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"{out}/{ctr}/**/*counter_collection.csv", recursive=True)
    if not f: print("missing", ctr); continue
    for r in csv.DictReader(open(f[0])):
        if r["Counter_Name"] != ctr: continue
        k = r["Kernel_Name"]
        agg[k][ctr] += float(r["Counter_Value"])
        if ctr == "FETCH_SIZE": agg[k]["n"] += 1
lines = [f"{'kernel':70s} {'launches':>8s} {'read_MB/launch(2x FETCH)':>26s} {'write_MB/launch':>16s}"]
for k, v in sorted(agg.items(), key=lambda kv: -(kv[1]['FETCH_SIZE'] + kv[1]['WRITE_SIZE'])):
    n = max(v["n"], 1)
    lines.append(f"{k[:70]:70s} {n:8d} {2*v['FETCH_SIZE']*1024/n/1e6:26.2f} {v['WRITE_SIZE']*1024/n/1e6:16.2f}")
import os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
lines.append(f"# kernel_source_sha16: {bench.kernel_source_sha16()}")
open(out + "/traffic_summary.txt", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
