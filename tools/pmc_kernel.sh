#!/bin/bash
# SQ accounting of the kernels whose name contains $1 over a few demo steps; remaining arguments are name=value runtime options
set -u
cd /tmp; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
pat=$1; shift
tag=$(echo "$pat $*" | tr ' =' '__')
out=$R/gpurun_out/pmc_kernel/$tag; rm -rf $out; mkdir -p $out
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES" \
           "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_MISC GRBM_GUI_ACTIVE" \
           "SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_LDS GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE"; do
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out/p$i -o p -- python3 $R/tools/step_once.py "$@" > $out/p$i.out 2>&1 || echo "pass $i failed: $(tail -2 $out/p$i.out)"
  i=$((i+1))
done
python3 - $out "$pat" <<'PY'
import csv, sys, glob, collections
out, pat = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(int)
for p in sorted(glob.glob(f"{out}/p*/")):
    for f in glob.glob(f"{p}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
            if pat not in k: continue
            k = k.replace("void mocha::", "")[:48]
            name = r["Counter_Name"]
            if name == "GRBM_GUI_ACTIVE": name = f"GUI@{p.rstrip('/').split('/')[-1]}"
            agg[k][name] += float(r["Counter_Value"]); n[(k, name)] += 1
for k, v in sorted(agg.items()):
    print(f"--- {k}")
    for name in sorted(v):
        c = n[(k, name)]
        print(f"   {name:36s} per launch {v[name] / c:16.1f}   (launches {c})")
PY
