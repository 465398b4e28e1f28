#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for v in 1 0 1 0; do echo "CC_OUTER=$v"; STEP_POOL_CC_OUTER=$v timeout 300 python bench.py --steps 100 --warmup 20 --no-cpu-baseline 2>/dev/null | cut -c1-110; done
for v in 1 0; do echo "c3 CC_OUTER=$v"; STEP_POOL_CC_OUTER=$v timeout 300 python bench.py --config c3 --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | cut -c1-110; done
