#!/bin/bash
# The per-GPU share of BASELINE configs[3] under the profiler: rocprofv3 --kernel-trace --stats of characterize(128 windows) against the
# 4 096-entry bf16 bank (true kernel durations; HIP-event pairs around single small kernels overstate them), and the HBM-side bytes of its
# kernels (separate --pmc FETCH_SIZE / WRITE_SIZE passes; read side x 2 on gfx950) -> gpurun_out/mid_trace/{stats,traffic}.txt
set -u
cd /tmp; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
W=${1:-128}
out=$R/gpurun_out/mid_trace; rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o t -- python3 $R/tools/mid_sites.py $W > $out/stats_stdout.txt 2> $out/stats_stderr.txt
f=$(find $out/stats -name '*kernel_stats.csv' | head -1)
python3 - "$f" "$out/stats.txt" "$W" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
lines = [f"# rocprofv3 --kernel-trace --stats -- python3 tools/mid_sites.py {sys.argv[3]}   (60 profiled-free steps + 5 with HIP events; set-up kernels included)",
         f"{'kernel':90s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>10s} {'pct':>6s}"]
for r in rows[:40]:
    lines.append(f"{r['Name'][:90]:90s} {r['Calls']:>7s} {float(r['TotalDurationNs'])/1e6:10.3f} {float(r['AverageNs'])/1e3:10.2f} {float(r['Percentage']):6.2f}")
open(sys.argv[2], "w").write("\n".join(lines) + "\n")
PY
for ctr in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $out/$ctr -o t -- python3 $R/tools/mid_sites.py $W > $out/$ctr.stdout 2> $out/$ctr.stderr
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
lines = [f"{'kernel':80s} {'launches':>8s} {'read_MB/launch(2x FETCH)':>26s} {'write_MB/launch':>16s}"]
for k, v in sorted(agg.items(), key=lambda kv: -(kv[1]['FETCH_SIZE'] + kv[1]['WRITE_SIZE'])):
    n = max(v["n"], 1)
    lines.append(f"{k[:80]:80s} {n:8d} {2*v['FETCH_SIZE']*1024/n/1e6:26.2f} {v['WRITE_SIZE']*1024/n/1e6:16.2f}")
open(out + "/traffic.txt", "w").write("\n".join(lines) + "\n")
print("\n".join(lines[:16]))
PY
