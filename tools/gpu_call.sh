#!/bin/bash
# one gpurun call: GPU tests, A/B of the two-phase conv form, headline bench per variant (outputs under gpurun_out/)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 1200 python -m pytest tests -m gpu -q > $O/gputests.log 2>&1; echo "rc=$?" >> $O/gputests.log
L3="2c_3x3,3b_b1b,3b_b2b,3c_b1b,3c_b2b,4b_b1b,4b_b2b,4c_b1b,4d_b1b,4e_b1b,4f_b1b,4f_b2b"
timeout 600 python tools/ab_bench.py --rounds 5 --iters 10 --only $L3 --var "STEP_CONV_PHASED=" --var "STEP_CONV_PHASED=1" --var "STEP_CONV_PHASED=2" > $O/ab_phased.log 2>&1
timeout 300 python bench.py --steps 50 --warmup 10 > $O/bench_c2.json 2> $O/bench_c2.err
STEP_CONV_PHASED=1 timeout 300 python bench.py --steps 50 --warmup 10 > $O/bench_c2_ph1.json 2> $O/bench_c2_ph1.err
STEP_CONV_PHASED=2 timeout 300 python bench.py --steps 50 --warmup 10 > $O/bench_c2_ph2.json 2> $O/bench_c2_ph2.err
tail -5 $O/gputests.log; cat $O/ab_phased.log | tail -16; for f in bench_c2 bench_c2_ph1 bench_c2_ph2; do cut -c1-120 $O/$f.json; done
