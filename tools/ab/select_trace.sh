#!/bin/bash
# true durations (rocprofv3 kernel trace) and wave-cycle accounting of the many-query selection kernels at 128 x 4096 bf16
set -u
cd /tmp; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/select_trace; rm -rf $out; mkdir -p $out
for sel in 0 1; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/t$sel -o t -- python3 $R/tools/ab/select_once.py 128 4096 $sel > $out/t$sel.out 2>&1
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $out/p$sel -o p -- python3 $R/tools/ab/select_once.py 128 4096 $sel > $out/p$sel.out 2>&1
  rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum --kernel-trace --output-format csv -d $out/c$sel -o c -- python3 $R/tools/ab/select_once.py 128 4096 $sel > $out/c$sel.out 2>&1
done
python3 - $out <<'PY'
import csv, sys, glob, collections
out = sys.argv[1]
for sel in (0, 1):
    f = glob.glob(f"{out}/t{sel}/**/*kernel_stats.csv", recursive=True)
    print(f"--- select2={sel}: kernel stats")
    for r in csv.DictReader(open(f[0])):
        if "match" in r["Name"] or "center" in r["Name"]:
            print(f"   {r['Name'][:70]:70s} calls {r['Calls']:>4s} avg {float(r['AverageNs'])/1e3:8.2f} us  min {float(r['MinNs'])/1e3:8.2f}  max {float(r['MaxNs'])/1e3:8.2f}")
    for tag in ("p", "c"):
        agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
        for f in glob.glob(f"{out}/{tag}{sel}/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                k = r["Kernel_Name"].split("(")[0][:40]
                if "select" not in k: continue
                agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
        for k, v in agg.items():
            print("   " + k + ": " + "  ".join(f"{c}={val / n[(k, c)]:.0f}" for c, val in sorted(v.items())))
PY
