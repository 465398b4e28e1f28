#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 300 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "row_groups or conv_units or golden_units" 2>&1 | tail -2
timeout 400 python tools/ab_bench.py --var "STEP_CONV_GMODE=0" --var "STEP_CONV_GMODE=1" 2>&1 | grep -E "layer|_3x3|b1b|b2b|total" > $O/ab_gmode.log; cat $O/ab_gmode.log
for v in 0 1 0 1; do echo "GMODE=$v"; STEP_CONV_GMODE=$v timeout 300 python bench.py --steps 100 --warmup 20 --no-cpu-baseline 2>/dev/null | cut -c1-110; done
