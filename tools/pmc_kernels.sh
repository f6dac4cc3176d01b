#!/bin/bash
# SQ counters for every kernel of a short bench run (diagnostic): MFMA busy, waits, LDS conflicts, clock
cd /tmp; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/pmc_kernels; rm -rf $out; mkdir -p $out
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $out -o k -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --sustained-s 0 > $out/stdout.txt 2> $out/stderr.txt
python3 - $out <<'PY'
import csv, sys, glob, collections
out = sys.argv[1]
cc = list(csv.DictReader(open(glob.glob(out + "/**/*counter_collection.csv", recursive=True)[0])))
kt = {r["Dispatch_Id"]: r for r in csv.DictReader(open(glob.glob(out + "/**/*kernel_trace.csv", recursive=True)[0]))}
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for r in cc:
    k = r["Kernel_Name"][:60]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
        t = kt[r["Dispatch_Id"]]; agg[k]["dur"] += (int(t["End_Timestamp"]) - int(t["Start_Timestamp"])) * 1e-9; agg[k]["n"] += 1
print(f"{'kernel':60s} {'n':>4s} {'us/launch':>9s} {'GHz':>5s} {'MFMA%':>6s} {'wait_any%':>9s} {'wait_inst%':>10s} {'active%':>8s} {'ldsconf/ldsact':>14s}")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1]["dur"]):
    if not v["dur"]: continue
    clk = v["GRBM_GUI_ACTIVE"] / 8 / v["dur"]
    mf = v["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / (clk * v["dur"]) * 100
    wc = v["SQ_WAVE_CYCLES"] or 1
    print(f"{k:60s} {int(v['n']):4d} {v['dur']/v['n']*1e6:9.1f} {clk/1e9:5.2f} {mf:6.1f} {v['SQ_WAIT_ANY']/wc*100:9.1f} {v['SQ_WAIT_INST_ANY']/wc*100:10.1f} {v['SQ_ACTIVE_INST_ANY']/wc*100:8.1f} {v['SQ_LDS_BANK_CONFLICT']/(v['SQ_ACTIVE_INST_LDS'] or 1):14.3f}")
PY
