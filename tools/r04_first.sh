#!/bin/bash
# round 4, first GPU pass: the GPU suite, the N = 1 bank4k checksums, the default line, a 2-rank stand-in default line
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; out=gpurun_out/r04a; mkdir -p $out
python3 -m pytest tests -q -m gpu -x -s -p no:cacheprovider > $out/pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> $out/pytest_gpu.txt
grep "\[structured\]" $out/pytest_gpu.txt > $out/structured.txt
tail -5 $out/pytest_gpu.txt
python3 bench.py --workload bank4k > $out/bank4k_v22.json 2> $out/bank4k_v22.err
python3 bench.py --workload bank4k --joints 24 > $out/bank4k_v24.json 2> $out/bank4k_v24.err
python3 bench.py > $out/bench.json 2> $out/bench.err
MOCHA_RCCL_LIBRARY=$R/tests/rccl_standin/librccl_standin.so MOCHA_BENCH_ONE_GPU=1 MOCHA_BENCH_BACKEND=gloo python3 bench.py --gpus 2 --no-cpu-baseline > $out/bench_2rank_standin.json 2> $out/bench_2rank_standin.err
head -c 400 $out/bank4k_v22.json; echo; head -c 600 $out/bench.json; echo
