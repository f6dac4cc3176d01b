#!/bin/bash
# builds tools/bin/match_pass_probe (the many-query coarse pass: round-4 kernel vs match_pass.hip variants, cold and warm, fill-only floors)
set -e
cd "$(dirname "$0")/.."
make -C mocha_sigasia2023_amd/csrc -j4 >/dev/null
mkdir -p tools/bin
H="/opt/rocm/bin/hipcc --offload-arch=gfx950"
C=mocha_sigasia2023_amd/csrc
$H -O3 -std=c++17 -w -I $C -c tools/match_pass_probe.hip -o tools/bin/match_pass_probe.o
$H tools/bin/match_pass_probe.o $C/match_mfma.o $C/match_pass.o -o tools/bin/match_pass_probe
