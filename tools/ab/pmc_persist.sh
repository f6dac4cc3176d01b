#!/bin/bash
# MFMA busy and wave-cycle accounting of the plane GEMM, persistent instance (768 workgroups) against the shipped one, over demo steps
set -u
cd /tmp; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/pmc_persist; rm -rf $out; mkdir -p $out
for pers in 0 768; do
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES --kernel-trace --output-format csv -d $out/p$pers -o p -- python3 $R/tools/ab/persist_once.py $pers > $out/p$pers.out 2>&1
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/t$pers -o t -- python3 $R/tools/ab/persist_once.py $pers > $out/t$pers.out 2>&1
done
python3 - $out <<'PY'
import csv, sys, glob, collections
out = sys.argv[1]
for pers in (0, 768):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for f in glob.glob(f"{out}/p{pers}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
            if "gemm_x3" not in k: continue
            k = k.replace("void mocha::", "")[:44]
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
    print(f"--- gemm_persistent={pers}")
    for k, v in sorted(agg.items()):
        c = n[(k, "GRBM_GUI_ACTIVE")]
        gui = v["GRBM_GUI_ACTIVE"] / 8            # summed over the 8 XCDs
        print(f"   {k:44s} launches {c:4d}  MFMA busy {v['SQ_VALU_MFMA_BUSY_CYCLES'] / (gui * 1024):.3f}  waves/SIMD {v['SQ_WAVE_CYCLES'] * 4 / (gui * 1024):.2f}  "
              f"parked {v['SQ_WAIT_ANY'] / v['SQ_WAVE_CYCLES']:.2f}  issue-stall {v['SQ_WAIT_INST_ANY'] / v['SQ_WAVE_CYCLES']:.2f}  issuing {v['SQ_ACTIVE_INST_ANY'] / v['SQ_WAVE_CYCLES']:.2f}")
    f = glob.glob(f"{out}/t{pers}/**/*kernel_stats.csv", recursive=True)
    for r in csv.DictReader(open(f[0])):
        if "gemm_x3" in r["Name"]:
            print(f"   trace: {r['Name'][:60]:60s} calls {r['Calls']:>4s} avg {float(r['AverageNs'])/1e3:8.2f} us total {float(r['TotalDurationNs'])/1e6:8.3f} ms")
PY
