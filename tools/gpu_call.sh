#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout 300 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "wgrad" 2>&1 | tail -2
timeout 300 python bench.py --config c4 --dtype bf16 --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | cut -c60-150
timeout 300 python bench.py --config c4 --dtype f32 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | cut -c60-150
