#!/bin/bash
# LDS bank-conflict share of the kernels whose name contains $1 over a few demo steps
set -u
cd /tmp; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
pat=$1; shift
out=$R/gpurun_out/pmc_lds; rm -rf $out; mkdir -p $out
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $out/p -o p -- python3 $R/tools/step_once.py "$@" > $out/p.out 2>&1 || tail -3 $out/p.out
python3 - $out "$pat" <<'PY'
import csv, sys, glob, collections
out, pat = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(int)
for f in glob.glob(f"{out}/p/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if pat not in k: continue
        agg[k.replace("void mocha::", "")[:50]][r["Counter_Name"]] += float(r["Counter_Value"]); n[k] += 1
for k, v in sorted(agg.items()):
    idx = max(v.get("SQ_LDS_IDX_ACTIVE", 0.0), 1.0)
    print(f"{k:52s} bank-conflict cycles / LDS active cycles {v.get('SQ_LDS_BANK_CONFLICT', 0) / idx:.3f}   addr-conflict {v.get('SQ_LDS_ADDR_CONFLICT', 0) / idx:.3f}   LDS active / GUI {idx / max(v.get('GRBM_GUI_ACTIVE', 1), 1) * 8 / 256:.3f} per CU")
PY
