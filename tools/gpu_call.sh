#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for v in 4 2 1; do echo "MINSEG=$v"; STEP_POOL_MINSEG=$v timeout 300 python tools/pool_bench.py 2>&1 | grep -E "3b_pool|3c_pool|pool4a|4b_pool|4c_pool|4f_pool|total"; done
