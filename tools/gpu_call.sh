#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 400 python tools/ab_bench.py --var "STEP_CONV_NB_SMALL=" --var "STEP_CONV_NB_SMALL=2" --var "STEP_CONV_NB_SMALL=3" 2>&1 | grep -E "layer|4._b1b|4._b2b|total"
for v in "" 2 3 "" 2; do echo "NB_SMALL=$v"; STEP_CONV_NB_SMALL=$v timeout 300 python bench.py --steps 100 --warmup 20 --no-cpu-baseline 2>/dev/null | cut -c1-110; done
