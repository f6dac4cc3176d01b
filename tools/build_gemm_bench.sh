#!/bin/bash
set -e
cd "$(dirname "$0")/.."
make -C mocha_sigasia2023_amd/csrc -j4 >/dev/null
mkdir -p tools/bin
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -w -I mocha_sigasia2023_amd/csrc -c tools/gemm_bench.hip -o tools/bin/gemm_bench.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 tools/bin/gemm_bench.o mocha_sigasia2023_amd/csrc/gemm_f32.o mocha_sigasia2023_amd/csrc/gemm_split.o -o tools/bin/gemm_bench
