#!/bin/bash
# One GPU pass that produces everything a round's profiles/ directory cites, from the CURRENT tree:
#   <tag>_bench.json                     the default command's line (python bench.py)
#   <tag>_kernel_stats_summary.txt       rocprofv3 --kernel-trace --stats of the SAME default command (its JSON line beside it)
#   <tag>_pmc_traffic_summary.txt        FETCH_SIZE / WRITE_SIZE per kernel (separate passes), stamped with the kernel sources' hash
#   <tag>_pmc_mfma_util.txt              SQ_VALU_MFMA_BUSY_CYCLES per SIMD cycle per kernel
#   <tag>_sites.txt, <tag>_stream_sites.txt   per-call-site HIP-event times of the demo step and of one streamed window
# usage (on the GPU box): bash tools/round_profile.sh <out_dir> <tag>
set -u
out=${1:-gpurun_out/prof}; tag=${2:-x}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd "$R"; mkdir -p "$out"; export TMPDIR=/tmp
python3 bench.py > "$out/${tag}_bench.json" 2> "$out/${tag}_bench.err"
tmp=gpurun_out/_rp_$tag; rm -rf "$tmp"; mkdir -p "$tmp"
rocprofv3 --kernel-trace --stats --output-format csv -d "$tmp/stats" -o bench -- python3 bench.py > "$out/${tag}_bench_under_rocprof.json" 2> "$tmp/stats_stderr.txt"
f=$(find "$tmp/stats" -name '*kernel_stats.csv' | head -1)
python3 - "$f" "$out/${tag}_kernel_stats_summary.txt" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
lines = ["# rocprofv3 --kernel-trace --stats -- python3 bench.py   (the default command)",
         f"{'kernel':90s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>10s} {'pct':>6s}"]
for r in rows:
    lines.append(f"{r['Name'][:90]:90s} {r['Calls']:>7s} {float(r['TotalDurationNs'])/1e6:10.3f} {float(r['AverageNs'])/1e3:10.2f} {float(r['Percentage']):6.2f}")
open(sys.argv[2], "w").write("\n".join(lines) + "\n")
PY
bash tools/pmc_traffic.sh "rp_$tag" > "$tmp/pmc_traffic.log" 2>&1; cp "gpurun_out/pmc_traffic_rp_$tag/traffic_summary.txt" "$out/${tag}_pmc_traffic_summary.txt"
bash tools/pmc_mfma_util.sh "rp_$tag" > "$tmp/pmc_mfma.log" 2>&1; cp "gpurun_out/pmc_mfma_rp_$tag/mfma_util_summary.txt" "$out/${tag}_pmc_mfma_util.txt"
python3 tools/profile_sites.py > "$out/${tag}_sites.txt" 2>&1
python3 tools/stream_sites.py > "$out/${tag}_stream_sites.txt" 2>&1
rm -rf "$tmp" "gpurun_out/pmc_traffic_rp_$tag" "gpurun_out/pmc_mfma_rp_$tag"
head -c 700 "$out/${tag}_bench.json"; echo; head -8 "$out/${tag}_kernel_stats_summary.txt"
