#!/bin/bash
# HBM-side bytes of one GEMM shape of tools/bin/gemm_bench (K sweep mode): FETCH_SIZE / WRITE_SIZE in separate passes.
# usage: tools/pmc_gemm_shape.sh <N>     (M = 1170*90 rows, K swept)
N=${1:-256}
cd /tmp; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/pmc_gemm_shape; rm -rf $out; mkdir -p $out
for ctr in FETCH_SIZE WRITE_SIZE; do
  MOCHA_BENCH_KSWEEP=$N rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $out/$ctr -o t -- $R/tools/bin/gemm_bench 1170 2 0 > $out/$ctr.stdout 2> $out/$ctr.stderr
done
python3 - $out <<'PY'
import csv, sys, glob, collections
out = sys.argv[1]
rows = {}
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"{out}/{ctr}/**/*counter_collection.csv", recursive=True)
    vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(f[0])) if r["Counter_Name"] == ctr and "gemm_f32" in r["Kernel_Name"]]
    rows[ctr] = vals
ks = [32, 64, 128, 256, 384, 512, 768, 1024, 1536, 2048]
per = len(rows["FETCH_SIZE"]) // len(ks)
print("K     read_MB(2xFETCH)  write_MB   algorithmic A / C (MB)")
for i, k in enumerate(ks):
    f = rows["FETCH_SIZE"][i * per:(i + 1) * per]; w = rows["WRITE_SIZE"][i * per:(i + 1) * per]
    print(f"{k:5d} {2*sum(f)/len(f)*1024/1e6:12.1f} {sum(w)/len(w)*1024/1e6:12.1f}      {105300*k*4/1e6:8.1f} / {105300*int(sys.argv[2]) * 4/1e6 if len(sys.argv)>2 else 0:8.1f}")
PY
