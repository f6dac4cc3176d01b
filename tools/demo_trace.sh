#!/bin/bash
# rocprofv3 kernel trace of examples/demo_pair.py -> gpurun_out/demo_trace
set -u
cd /tmp; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/demo_trace; rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out/raw -o t -- python3 $R/examples/demo_pair.py --frames 585 --out $out/bvh > $out/stdout.txt 2> $out/stderr.txt
f=$(find $out/raw -name '*kernel_stats.csv' | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
print(f"{'kernel':80s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>9s} {'pct':>6s}")
for r in rows[:40]:
    print(f"{r['Name'][:80]:80s} {r['Calls']:>7s} {float(r['TotalDurationNs'])/1e6:10.3f} {float(r['AverageNs'])/1e3:9.2f} {float(r['Percentage']):6.2f}")
PY
tail -3 $out/stdout.txt
rm -rf $out/raw $out/bvh
