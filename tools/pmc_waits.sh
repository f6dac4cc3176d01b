#!/bin/bash
# Where the waves of each kernel of the bench step spend their cycles: parked (s_waitcnt / barrier), issue-stalled, issuing; LDS
# bank conflicts.  One PMC pass (8 SQ counters), kernel-trace only.
tag=${1:-r02}
cd /tmp; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/pmc_waits_$tag; rm -rf $out; mkdir -p $out
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d $out -o w -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --sustained-s 0 > $out/stdout.txt 2> $out/stderr.txt
python3 - $out <<'PY'
import csv, sys, glob, collections
out = sys.argv[1]
f = glob.glob(out + "/**/*counter_collection.csv", recursive=True)
if not f: print("no counter file"); sys.exit(1)
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(f[0])):
    agg[r["Kernel_Name"]][r["Counter_Name"]] += float(r["Counter_Value"])
lines = [f"{'kernel':60s} {'parked':>8s} {'issue-stall':>12s} {'issuing':>8s} {'wait LDS':>9s} {'LDS conflict':>13s}"]
for k, v in sorted(agg.items(), key=lambda kv: -kv[1]["SQ_WAVE_CYCLES"])[:12]:
    wc = v["SQ_WAVE_CYCLES"] or 1
    lines.append(f"{k[:60]:60s} {v['SQ_WAIT_ANY']/wc:8.3f} {v['SQ_WAIT_INST_ANY']/wc:12.3f} {v['SQ_ACTIVE_INST_ANY']/wc:8.3f} {v['SQ_WAIT_INST_LDS']/wc:9.3f} {v['SQ_LDS_BANK_CONFLICT']/max(v['SQ_LDS_IDX_ACTIVE'],1):13.3f}")
open(out + "/waits_summary.txt", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
