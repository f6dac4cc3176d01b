#!/bin/bash
# rocprofv3 kernel trace of the streamed window (tools/stream_bench.py, 16 384-row bank): true per-kernel durations of the 40-launch
# B = 1 chain (HIP-event pairs around single small kernels overstate them) -> gpurun_out/stream_trace/summary.txt
cd /tmp; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/stream_trace; rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out/raw -o t -- python3 $R/tools/stream_bench.py > $out/bench.txt 2> $out/stderr.txt
f=$(find $out/raw -name '*kernel_stats.csv' | head -1)
python3 - "$f" "$out/summary.txt" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
lines = ["# rocprofv3 --kernel-trace --stats -- python3 tools/stream_bench.py  (4 x 295 streamed windows: fp32 / bf16 bank, eager / graph)",
         f"{'kernel':90s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>9s} {'min_us':>8s} {'pct':>6s}"]
tot = 0.0
for r in rows:
    lines.append(f"{r['Name'][:90]:90s} {r['Calls']:>7s} {float(r['TotalDurationNs'])/1e6:10.3f} {float(r['AverageNs'])/1e3:9.2f} {float(r['MinNs'])/1e3:8.2f} {float(r['Percentage']):6.2f}")
    tot += float(r['TotalDurationNs'])
lines.append(f"# sum of kernel durations {tot/1e6:.1f} ms")
open(sys.argv[2], "w").write("\n".join(lines) + "\n")
print("\n".join(lines[:40]))
PY
grep "bank=" $out/bench.txt
rm -rf $out/raw
