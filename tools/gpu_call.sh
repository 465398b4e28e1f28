#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 600 python tools/wgrad_bench.py --dtype bf16 > $O/wgrad_bench.log 2>&1
timeout 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "wgrad" > $O/gputests.log 2>&1; echo "rc=$?" >> $O/gputests.log
timeout 300 python bench.py --config c4 --dtype bf16 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_c4_bf16.json 2> $O/bench_c4_bf16.err
cat $O/wgrad_bench.log; tail -3 $O/gputests.log; cut -c1-160 $O/bench_c4_bf16.json
