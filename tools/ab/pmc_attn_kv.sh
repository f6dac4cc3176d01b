#!/bin/bash
# PMC passes over the decoder at 585 windows, attention from key / value images (1) and from the fp32 rows (0): where the waves'
# cycles go, LDS conflicts, MFMA busy; texture-addresser / L1 / L2 request counts.  -> gpurun_out/pmc_attn_kv/summary.txt
set -u
cd /tmp; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/pmc_attn_kv; rm -rf $out; mkdir -p $out
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE" \
           "TA_BUSY_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum"; do
  i=$((i+1))
  for kv in 0 1; do
    rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out/p${i}_kv$kv -o w -- python3 $R/tools/ab/decoder_once.py 585 $kv > $out/p${i}_kv$kv.stdout 2> $out/p${i}_kv$kv.stderr
  done
done
python3 - $out <<'PY'
import csv, sys, glob, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter()
for f in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
    seen = set()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "attention_x3" not in k and "instnorm" not in k: continue
        k = k.split("(")[0][:48]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        key = (k, r["Dispatch_Id"], r["Counter_Name"])
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "attention_x3" not in k and "instnorm" not in k: continue
        k = k.split("(")[0][:48]
        if r["Counter_Name"] in ("SQ_WAVE_CYCLES", "SQ_INSTS_VALU", "TA_BUSY_sum", "TCC_HIT_sum"): calls[(k, r["Counter_Name"])] += 1
lines = []
for k, v in sorted(agg.items()):
    lines.append(k)
    for c, val in sorted(v.items()):
        n = max([calls[(k, cc)] for cc in ("SQ_WAVE_CYCLES", "SQ_INSTS_VALU", "TA_BUSY_sum", "TCC_HIT_sum") if calls[(k, cc)]] or [1])
        lines.append(f"    {c:36s} {val / n:16.0f} per launch")
open(out + "/summary.txt", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
