#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 300 python tools/ab_bench.py --set b3 --rounds 5 --iters 10 > $O/ab_b3.log 2>&1
timeout 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "pool" > $O/gputests.log 2>&1; echo "rc=$?" >> $O/gputests.log
for i in 1 2; do
STEP_FUSE_POOL_CONV=0 timeout 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline > $O/bench_c2_nofuse$i.json 2> $O/bench_c2_nofuse.err
STEP_FUSE_POOL_CONV=1 timeout 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline > $O/bench_c2_fuse$i.json 2> $O/bench_c2_fuse.err
timeout 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline > $O/bench_c2_$i.json 2> $O/bench_c2.err
done
cat $O/ab_b3.log; tail -3 $O/gputests.log; for f in bench_c2_nofuse1 bench_c2_fuse1 bench_c2_1 bench_c2_nofuse2 bench_c2_fuse2 bench_c2_2; do echo $f $(cut -c1-110 $O/$f.json); done
