#!/bin/bash
# Kernel-trace + stats profile of the default bench command (run on the GPU box via gpurun).
# usage: tools/rocprof_bench.sh <tag>   -> gpurun_out/prof_<tag>/ and a text summary
set -u
tag=${1:-r01}
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
out=gpurun_out/prof_$tag
rm -rf "$out"; mkdir -p "$out"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -o bench -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --sustained-s 0 ${BENCH_EXTRA:---no-extras} > "$out/bench_stdout.json" 2> "$out/bench_stderr.txt"
f=$(find "$out" -name '*kernel_stats.csv' | head -1)
echo "stats file: $f"
python3 - "$f" "$out" <<'PY'
import csv, sys
f, out = sys.argv[1], sys.argv[2]
rows = list(csv.DictReader(open(f)))
lines = [f"{'kernel':90s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>10s} {'pct':>6s}"]
for r in rows:
    lines.append(f"{r['Name'][:90]:90s} {r['Calls']:>7s} {float(r['TotalDurationNs'])/1e6:10.3f} {float(r['AverageNs'])/1e3:10.2f} {float(r['Percentage']):6.2f}")
open(out + "/kernel_stats_summary.txt", "w").write("\n".join(lines) + "\n")
print("\n".join(lines[:25]))
PY
