#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
prof() { n=$1; shift
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$n -- python $R/bench.py "$@" > $O/bench_${n}_prof.json 2> $O/bench_${n}_prof.err
  python $R/tools/prof_summary.py $O/prof_$n $O/prof_${n}_summary.txt > /dev/null 2>&1
  rm -rf $O/prof_$n; }
prof c4_bf16 --config c4 --dtype bf16 --steps 10 --warmup 3 --no-cpu-baseline
head -36 $O/prof_c4_bf16_summary.txt | cut -c1-150
