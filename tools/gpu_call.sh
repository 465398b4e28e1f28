#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
for v in 2 1 0 2 1; do echo "STEP_BRANCH_STREAMS=$v"; STEP_BRANCH_STREAMS=$v timeout 300 python bench.py --steps 100 --warmup 20 --no-cpu-baseline 2>/dev/null | cut -c1-110; done
