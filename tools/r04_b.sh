#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; out=gpurun_out/r04b; mkdir -p $out
python3 -m pytest tests/test_gemm_engines.py tests/test_hip_parity.py tests/test_structured_inputs.py tests/test_fullsize_parity.py -q -m gpu -x -s -p no:cacheprovider > $out/pytest.txt 2>&1; echo "pytest rc=$?" >> $out/pytest.txt
grep "\[structured\]\|\[attention_kv\]\|passed\|failed\|rc=" $out/pytest.txt > $out/summary.txt
python3 tools/attn_kv_ab.py > $out/attn_kv_ab.txt 2>&1
python3 bench.py --no-extras --no-cpu-baseline > $out/bench_noextras.json 2> $out/bench.err; python3 bench.py --no-extras --no-cpu-baseline > $out/bench_noextras2.json 2>> $out/bench.err
cat $out/summary.txt; cat $out/attn_kv_ab.txt; head -c 300 $out/bench_noextras.json
