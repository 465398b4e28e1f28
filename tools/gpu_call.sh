#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "two_phase or conv_units or big_conv" > $O/gputests.log 2>&1; echo "rc=$?" >> $O/gputests.log
L3="2c_3x3,3b_b1b,3b_b2b,3c_b1b,3c_b2b,4b_b1b,4c_b1b,4d_b1b,4e_b1b,4f_b1b,4f_b2b"
timeout 600 python tools/ab_bench.py --rounds 5 --iters 10 --only $L3 --var "STEP_CONV_PHASED=" --var "STEP_CONV_PHASED=1" > $O/ab_phased.log 2>&1
STEP_CONV_PHASED=1 timeout 300 python bench.py --steps 50 --warmup 10 > $O/bench_c2_ph1.json 2> $O/bench_c2_ph1.err
tail -3 $O/gputests.log; cat $O/ab_phased.log | tail -14; cut -c1-120 $O/bench_c2_ph1.json
