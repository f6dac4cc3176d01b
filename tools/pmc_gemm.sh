#!/bin/bash
# PMC counters for the GEMM micro-benchmark (counters in their own run; kernel-trace only)
# usage: pmc_gemm.sh [mode 0|36] [windows] [binary in tools/bin]
cd /tmp; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/pmc_gemm_${1:-0}_${3:-gemm_bench}; rm -rf $out; mkdir -p $out
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $out -o g -- $R/tools/bin/${3:-gemm_bench} ${2:-585} 1 0 ${1:-0} > $out/stdout.txt 2>&1
python3 - $out <<'PY'
import csv, sys, glob, collections
out = sys.argv[1]
f = glob.glob(out + "/**/*counter_collection.csv", recursive=True)
if not f: print("no counter file", glob.glob(out+"/**/*", recursive=True)); sys.exit()
rows = list(csv.DictReader(open(f[0])))
agg = collections.OrderedDict()
for r in rows:
    if "mocha_gemm" not in r["Kernel_Name"] or "skinny" in r["Kernel_Name"]: continue
    key = r["Dispatch_Id"]
    agg.setdefault(key, {})[r["Counter_Name"]] = float(r["Counter_Value"])
    agg[key]["grid"] = r.get("Grid_Size")
names = ["SQ_WAVE_CYCLES","SQ_BUSY_CYCLES","SQ_WAIT_ANY","SQ_WAIT_INST_ANY","SQ_ACTIVE_INST_ANY","SQ_VALU_MFMA_BUSY_CYCLES","SQ_ACTIVE_INST_LDS","SQ_LDS_BANK_CONFLICT"]
print("disp grid " + " ".join(n[3:] for n in names))
for k, v in list(agg.items()):
    print(k, v.get("grid"), " ".join(f"{v.get(n,0):.3g}" for n in names))
PY
