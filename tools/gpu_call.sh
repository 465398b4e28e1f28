#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout 300 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "wgrad" 2>&1 | tail -2
for v in 1 0; do echo "NOREMAP=$v"; STEP_WGRAD16_NOREMAP=$v timeout 300 python tools/wgrad_bench.py --dtype bf16 2>&1 | grep -v amdgpu | cut -c1-110; done
for v in 1 0 1 0; do echo "NOREMAP=$v"; STEP_WGRAD16_NOREMAP=$v timeout 300 python bench.py --config c4 --dtype bf16 --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | cut -c60-150; done
