#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; out=gpurun_out/r04d; mkdir -p $out
python3 -m pytest tests/test_edge_cases.py tests/test_hip_parity.py tests/test_fullsize_parity.py tests/test_runtime_gpu.py tests/test_bank.py -q -m gpu -x -p no:cacheprovider > $out/pytest.txt 2>&1; echo "pytest rc=$?" >> $out/pytest.txt
tail -4 $out/pytest.txt
python3 tools/select_ab.py > $out/select_ab.txt 2>&1; cat $out/select_ab.txt
python3 bench.py --no-cpu-baseline > $out/bench.json 2> $out/bench.err
python3 - <<'PY'
import json
d = json.load(open("gpurun_out/r04d/bench.json"))
print(d["value"], d["ms_per_step"], d["match_sites"])
for k, v in d["match"].items(): print(k, round(v["us"], 1), round(v["frac_of_hbm_peak"], 3), {a: round(b, 1) for a, b in v["kernels"].items()})
PY
