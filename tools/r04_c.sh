#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; out=gpurun_out/r04c; mkdir -p $out
python3 -m pytest tests/test_gemm_engines.py tests/test_structured_inputs.py -q -m gpu -x -s -p no:cacheprovider > $out/pytest.txt 2>&1; echo "pytest rc=$?" >> $out/pytest.txt
grep "\[structured\]\|\[attention_kv\]\|passed\|failed\|rc=" $out/pytest.txt > $out/summary.txt
python3 tools/attn_kv_ab.py > $out/attn_kv_ab.txt 2>&1
cat $out/summary.txt; grep "B=  585\|B=  256\|B= 1024" $out/attn_kv_ab.txt
