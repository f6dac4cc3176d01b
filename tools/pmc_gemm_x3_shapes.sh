#!/bin/bash
# HBM-side bytes of mocha_gemm_x3 per SHAPE (the transformer's four launches by default): FETCH_SIZE / WRITE_SIZE in separate passes
# over tools/bin/gemm_bench (mode 36), grouped by launch order.  usage: tools/pmc_gemm_x3_shapes.sh [binary] ["M,N,K;M,N,K;..."]
BIN=${1:-gemm_bench}
SH=${2:-"105300,768,256;105300,256,256;105300,512,256;105300,256,512;52650,512,256"}
cd /tmp; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/pmc_x3_shapes_$BIN; rm -rf $out; mkdir -p $out
export MOCHA_BENCH_SHAPES="$SH"
for ctr in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $out/$ctr -o t -- $R/tools/bin/$BIN 1170 5 0 36 > $out/$ctr.stdout 2> $out/$ctr.stderr
done
cat $out/FETCH_SIZE.stdout
python3 - $out "$SH" <<'PY'
import csv, sys, glob
out, shapes = sys.argv[1], [tuple(int(v) for v in s.split(",")) for s in sys.argv[2].split(";") if s]
rows = {}
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"{out}/{ctr}/**/*counter_collection.csv", recursive=True)
    rs = [r for r in csv.DictReader(open(f[0])) if r["Counter_Name"] == ctr and "mocha_gemm_x3<" in r["Kernel_Name"]]
    rs.sort(key=lambda r: int(r["Dispatch_Id"]))
    rows[ctr] = [float(r["Counter_Value"]) for r in rs]
per = len(rows["FETCH_SIZE"]) // len(shapes)
print("M       N     K     read_MB(2xFETCH)  write_MB   algorithmic A / C (MB)   read / A")
for i, (m, n, k) in enumerate(shapes):
    f = rows["FETCH_SIZE"][i * per:(i + 1) * per]; w = rows["WRITE_SIZE"][i * per:(i + 1) * per]
    rd = 2 * sum(f) / len(f) * 1024 / 1e6
    print(f"{m:7d} {n:5d} {k:5d} {rd:12.1f} {sum(w)/len(w)*1024/1e6:12.1f}      {m*k*4/1e6:8.1f} / {m*n*4/1e6:8.1f}     {rd/(m*k*4/1e6):.2f}")
PY
