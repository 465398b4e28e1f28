#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
for c in c3 c5; do for v in 1 2 1 2; do echo "$c BRANCH_STREAMS=$v"; STEP_BRANCH_STREAMS=$v timeout 300 python bench.py --config $c --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | cut -c1-120; done; done
