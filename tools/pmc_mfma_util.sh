#!/bin/bash
# MFMA utilisation of every kernel of the default bench step, one PMC pass, kernel-trace only (BENCH_EXTRA="--options gemm_f16x2=1": with context options).
# On this chip rocprofv3 sums GRBM_GUI_ACTIVE over the 8 XCDs and the SQ counters over the 32 shader engines (SQ_BUSY_CYCLES /
# GRBM_GUI_ACTIVE = 4.0 for a kernel that keeps every engine busy), and SQ_VALU_MFMA_BUSY_CYCLES counts busy cycles per SIMD,
# so utilisation = MFMA_BUSY / (GUI_ACTIVE / 8) / 1024 SIMDs.
tag=${1:-r01}
cd /tmp; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/pmc_mfma_$tag; rm -rf $out; mkdir -p $out
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $out -o m -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --sustained-s 0 ${BENCH_EXTRA:-} > $out/stdout.txt 2> $out/stderr.txt
python3 - $out <<'PY'
import csv, sys, glob, collections
out = sys.argv[1]
f = glob.glob(out + "/**/*counter_collection.csv", recursive=True)
if not f: print("no counter file"); sys.exit(1)
agg = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.Counter()
for r in csv.DictReader(open(f[0])):
    k = r["Kernel_Name"]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "GRBM_GUI_ACTIVE": n[k] += 1
lines = [f"{'kernel':70s} {'launches':>8s} {'MFMA utilisation':>18s} {'engines busy':>14s}"]
for k, v in sorted(agg.items(), key=lambda kv: -kv[1]["GRBM_GUI_ACTIVE"]):
    gui = v["GRBM_GUI_ACTIVE"]
    if gui <= 0: continue
    lines.append(f"{k[:70]:70s} {n[k]:8d} {v['SQ_VALU_MFMA_BUSY_CYCLES'] / (gui / 8) / 1024:18.3f} {v['SQ_BUSY_CYCLES'] / gui / 4:14.3f}")
open(out + "/mfma_util_summary.txt", "w").write("\n".join(lines) + "\n")
print("\n".join(lines[:14]))
PY
