#!/bin/bash
# builds tools/bin/gemm_bench, tools/bin/gemm_bench_f16 / gemm_bench_f16k32 (the two-plane fp16 experiments, tools/experiments/gemm_f16x2_r05.hip / gemm_f16x2_k32_r05.hip, in place of gemm_x3.hip) (and, with an argument, an ablation of gemm_x3.hip: build_gemm_bench.sh -DX3_MAXSUM=0 -> tools/bin/gemm_bench_abl)
set -e
cd "$(dirname "$0")/.."
make -C mocha_sigasia2023_amd/csrc -j4 >/dev/null
mkdir -p tools/bin
H="/opt/rocm/bin/hipcc --offload-arch=gfx950"
$H -O3 -std=c++17 -w -I mocha_sigasia2023_amd/csrc -c tools/gemm_bench.hip -o tools/bin/gemm_bench.o
C=mocha_sigasia2023_amd/csrc
$H tools/bin/gemm_bench.o $C/gemm_f32.o $C/gemm_x3.o $C/gemm_h2.o $C/gemm_x3r.o -o tools/bin/gemm_bench
$H -O3 -fPIC -std=c++17 -w -I $C -c tools/experiments/gemm_f16x2_r05.hip -o tools/bin/gemm_f16x2.o
$H tools/bin/gemm_bench.o $C/gemm_f32.o tools/bin/gemm_f16x2.o $C/gemm_h2.o $C/gemm_x3r.o -o tools/bin/gemm_bench_f16
$H -O3 -fPIC -std=c++17 -w -I $C -c tools/experiments/gemm_f16x2_k32_r05.hip -o tools/bin/gemm_f16x2_k32.o
$H tools/bin/gemm_bench.o $C/gemm_f32.o tools/bin/gemm_f16x2_k32.o $C/gemm_h2.o $C/gemm_x3r.o -o tools/bin/gemm_bench_f16k32
# variants of the register-resident instance (gemm_x3r.hip): ring depth, diagnostics
for v in "r4:-DX3R_RING=4" "contig:-DX3R_CONTIGUOUS" "nopf:-DX3R_NOPREFETCH" "st:-DX3R_STAMPS"; do
  n=${v%%:*}; f=${v#*:}
  $H -O3 -fPIC -std=c++17 -w $f -c $C/gemm_x3r.hip -o tools/bin/gemm_x3r_$n.o
  $H tools/bin/gemm_bench.o $C/gemm_f32.o $C/gemm_x3.o $C/gemm_h2.o tools/bin/gemm_x3r_$n.o -o tools/bin/gemm_bench_x3r_$n
done
if [ -n "$1" ]; then
  $H -O3 -fPIC -std=c++17 -w "$@" -c $C/gemm_x3.hip -o tools/bin/gemm_x3_abl.o
  $H tools/bin/gemm_bench.o $C/gemm_f32.o tools/bin/gemm_x3_abl.o $C/gemm_h2.o $C/gemm_x3r.o -o tools/bin/gemm_bench_abl
fi
