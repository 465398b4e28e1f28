#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "stem_wgrad" 2>&1 | tail -3
timeout 300 python tools/stem_wgrad_bench.py 2>&1 | tail -2
timeout 300 python tools/stem_wgrad_bench.py --n 4 --t 32 --hw 224 2>&1 | tail -2
timeout 400 python bench.py --config c4 --dtype bf16 --steps 20 --warmup 5 --no-cpu-baseline 2>$O/bench_c4_bf16.err | tee $O/bench_c4_bf16.json | cut -c1-140
