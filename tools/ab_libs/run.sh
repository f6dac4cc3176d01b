#!/bin/bash
# alternate two builds of the library under the per-site profile of the demo step
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for rep in 1 2; do
  for v in old new; do
    cp tools/ab_libs/$v.so mocha_sigasia2023_amd/libmocha_hip.so
    echo "=== $v ($rep)"
    python3 tools/profile_sites.py 2>&1 | grep "ms/step\|gemm_x3" | cut -c1-110
  done
done
