#!/bin/bash
# rocprofv3 kernel trace of the CVAE ("Ours") branch per frame (tools/ours_bench.py): true per-kernel durations -> gpurun_out/ours_trace/summary.txt
cd /tmp; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/ours_trace; rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out/raw -o t -- python3 $R/tools/ours_bench.py > $out/bench.txt 2> $out/stderr.txt
f=$(find $out/raw -name '*kernel_stats.csv' | head -1)
python3 - "$f" "$out/summary.txt" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
lines = ["# rocprofv3 --kernel-trace --stats -- python3 tools/ours_bench.py  (820 frames: 1 and 8 clips, eager and graph)",
         f"{'kernel':90s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>9s} {'min_us':>8s} {'pct':>6s}"]
for r in rows:
    lines.append(f"{r['Name'][:90]:90s} {r['Calls']:>7s} {float(r['TotalDurationNs'])/1e6:10.3f} {float(r['AverageNs'])/1e3:9.2f} {float(r['MinNs'])/1e3:8.2f} {float(r['Percentage']):6.2f}")
open(sys.argv[2], "w").write("\n".join(lines) + "\n")
print("\n".join(lines[:32]))
PY
grep "clips=" $out/bench.txt
rm -rf $out/raw
