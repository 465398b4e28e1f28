#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -q > $O/gputests.log 2>&1; echo "rc=$?" >> $O/gputests.log
cd /tmp; export TMPDIR=/tmp
prof() { n=$1; shift
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$n -- python $R/bench.py "$@" --no-cpu-baseline > $O/bench_${n}_prof.json 2> $O/bench_${n}_prof.err
  python $R/tools/prof_summary.py $O/prof_$n $O/prof_${n}_summary.txt > /dev/null 2>&1
  find $O/prof_$n -type f ! -name "*kernel_stats*" -delete 2>/dev/null; }
prof c4_bf16 --config c4 --dtype bf16 --steps 10 --warmup 3
cd $R
tail -4 $O/gputests.log; head -30 $O/prof_c4_bf16_summary.txt | cut -c1-210
