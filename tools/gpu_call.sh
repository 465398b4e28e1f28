#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_modules.py tests/test_gpu_graph_step.py -q -m gpu -k "pool or backward or training or graph or chain" 2>&1 | tail -4
for v in 1 0 1 0; do echo "GATHER=$v"; STEP_POOL_BWD_GATHER=$v timeout 300 python bench.py --config c4 --dtype bf16 --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | cut -c60-150; done
for v in 1 0; do echo "f32 GATHER=$v"; STEP_POOL_BWD_GATHER=$v timeout 300 python bench.py --config c4 --dtype f32 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | cut -c60-150; done
