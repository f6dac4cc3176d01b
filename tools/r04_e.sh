#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; out=gpurun_out/r04e; mkdir -p $out
python3 -m pytest tests -q -m gpu -x -p no:cacheprovider > $out/pytest.txt 2>&1; echo "pytest rc=$?" >> $out/pytest.txt
tail -4 $out/pytest.txt
python3 bench.py > $out/bench.json 2> $out/bench.err
python3 - <<'PY'
import json
d = json.load(open("gpurun_out/r04e/bench.json"))
print(d["value"], d["ms_per_step"], d["match_sites"], d["cpu_baseline"]["value"])
for k, v in d["match"].items(): print(k, round(v["us"], 1), round(v["frac_of_hbm_peak"], 3), {a: round(b, 1) for a, b in v["kernels"].items()})
PY
