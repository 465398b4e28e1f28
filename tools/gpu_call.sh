#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_modules.py -q -m gpu -k "tail_round or conv_units or c2_full_size or c5_full or big_conv" 2>&1 | tail -2
timeout 400 python tools/ab_bench.py --var "STEP_CONV_TAIL=0" --var "STEP_CONV_TAIL=1" 2>&1 | grep -E "layer|2c_3x3|3c_b1b|3b_b1b|total"
timeout 400 python tools/ab_bench.py --set c3 --batch 4 --var "STEP_CONV_TAIL=0" --var "STEP_CONV_TAIL=1" 2>&1 | grep -E "layer|3x3|b1b|total"
for v in 0 1 0 1; do echo "TAIL=$v"; STEP_CONV_TAIL=$v timeout 300 python bench.py --steps 100 --warmup 20 --no-cpu-baseline 2>/dev/null | cut -c1-110; done
