#!/bin/bash
# round 4, GPU call 8: kernel statistics of the training step as it stands (8 clips x 15 tubes and 1 clip x 5 tubes, eager under rocprofv3)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
prof() { n=$1; shift
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$n -- python $R/bench.py "$@" > $O/r04f_${n}_prof.json 2> $O/r04f_${n}_prof.err
  python $R/tools/prof_summary.py $O/prof_$n $O/r04f_${n}_kernel_stats.txt > /dev/null 2>&1
  rm -rf $O/prof_$n; }
prof c4_bf16_b8 --config c4 --dtype bf16 --clips 8 --tubes 15 --steps 6 --warmup 2 --no-cpu-baseline --no-graph
prof c4_bf16 --config c4 --dtype bf16 --steps 10 --warmup 3 --no-cpu-baseline --no-graph
head -45 $O/r04f_c4_bf16_b8_kernel_stats.txt | cut -c1-150
