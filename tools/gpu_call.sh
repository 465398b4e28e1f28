#!/bin/bash
# one gpurun call: GPU tests, headline bench, A/B of the planner variants, rocprof kernel stats (outputs under gpurun_out/)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 1200 python -m pytest tests -m gpu -q > $O/gputests.log 2>&1; echo "rc=$?" >> $O/gputests.log
timeout 300 python bench.py --steps 50 --warmup 10 > $O/bench_c2.json 2> $O/bench_c2.err
STEP_FUSE_POOL_CONV=0 timeout 300 python bench.py --steps 50 --warmup 10 > $O/bench_c2_nofuse.json 2> $O/bench_c2_nofuse.err
timeout 300 python tools/ab_bench.py --set b3 --rounds 5 --iters 10 > $O/ab_b3.log 2>&1
timeout 600 python tools/ab_bench.py --rounds 3 --iters 10 --var "STEP_CONV_WAVES=" --var "STEP_CONV_WAVES=8" > $O/ab_plan.log 2>&1
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c2 -- python $R/bench.py --steps 50 --warmup 5 > $O/bench_c2_prof.json 2> $O/bench_c2_prof.err
cd $R
python tools/prof_summary.py $O/prof_c2 $O/prof_c2_summary.txt > /dev/null 2>&1
find $O/prof_c2 -type f ! -name "*kernel_stats*" -delete 2>/dev/null
tail -5 $O/gputests.log; cat $O/bench_c2.json; cat $O/ab_b3.log | tail -12
