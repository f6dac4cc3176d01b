#!/bin/bash
# effective shader clock during the GEMM micro-benchmark: GRBM_GUI_ACTIVE / 8 / kernel duration
# usage: pmc_clock.sh [mode 0|36] [windows]   (long dispatches - several thousand windows - give a trustworthy quotient)
cd /tmp; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/pmc_clock_${1:-0}; rm -rf $out; mkdir -p $out
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d $out -o g -- $R/tools/bin/gemm_bench ${2:-585} 1 0 ${1:-0} > $out/stdout.txt 2>&1
python3 - $out <<'PY'
import csv, sys, glob, collections
out = sys.argv[1]
cc = list(csv.DictReader(open(glob.glob(out + "/**/*counter_collection.csv", recursive=True)[0])))
kt = {r["Dispatch_Id"]: r for r in csv.DictReader(open(glob.glob(out + "/**/*kernel_trace.csv", recursive=True)[0]))}
agg = collections.OrderedDict()
for r in cc:
    if "mocha_gemm" not in r["Kernel_Name"]: continue
    agg.setdefault(r["Dispatch_Id"], {})[r["Counter_Name"]] = float(r["Counter_Value"])
    agg[r["Dispatch_Id"]]["grid"] = r["Grid_Size"]; agg[r["Dispatch_Id"]]["name"] = r["Kernel_Name"].split("(")[0][-28:]
for k, v in agg.items():
    t = kt[k]; dur = (int(t["End_Timestamp"]) - int(t["Start_Timestamp"])) * 1e-9
    clk = v.get("GRBM_GUI_ACTIVE", 0) / 8 / dur / 1e9
    busy = v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / 1024 / (clk * 1e9 * dur) if clk else 0
    print(f"disp {k:>3s} {v['name']:28s} grid {v['grid']:>9s} dur {dur*1e6:8.1f} us  clock {clk:5.2f} GHz  MFMA-busy/SIMD-cycles {busy*100:5.1f}%")
PY
