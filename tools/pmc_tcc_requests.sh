#!/bin/bash
# TCC request counters of the default bench command -> gpurun_out/tcc
set -u
cd /tmp; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/tcc; mkdir -p $out
rocprofv3 --list-avail 2>/dev/null | grep -o "TCC_EA0_[A-Z0-9_]*\|TCC_HIT[a-z_]*\|TCC_MISS[a-z_]*\|TCC_REQ[a-z_]*\|TCC_READ[a-z_]*\|TCC_WRITE[a-z_]*\|TCC_EA_[A-Z0-9_]*" | sort -u | tr '\n' ' ' > $out/avail.txt
cat $out/avail.txt; echo
for set in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum TCC_READ_sum"; do
  tag=$(echo $set | tr ' ' '+')
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out/$tag -o t -- python3 $R/tools/enc_once.py > $out/$tag.log 2>&1
  python3 - $out/$tag <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)
if not f: print("no csv for", sys.argv[1]); sys.exit()
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(f[0])):
    k = r["Kernel_Name"][:48]; agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
for k, v in agg.items():
    if "attention" in k or "gemm_x3" in k:
        print(k, {c: round(x / n[(k, c)]) for c, x in v.items()}, "launches", max(n[(k, c)] for c in v))
PY
done
