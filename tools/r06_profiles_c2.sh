#!/bin/bash
# round 6: only the C2 rocprofv3 kernel statistics + replayed-step timeline (bf16 only: --no-fp16-leg), one batch at a time and the default command
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
prof() { n=$1; shift
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$n -- python $R/bench.py "$@" 2> $O/r06p_${n}_prof.err | grep '^{' > $O/r06p_${n}_prof.json
  python $R/tools/prof_summary.py $O/prof_$n $O/r06p_${n}_kernel_stats.txt > /dev/null 2>&1
  if [ "$n" = "c2" ]; then python $R/tools/graph_timeline.py $O/prof_$n > $O/r06p_c2_graph_timeline.txt 2>&1; fi
  rm -rf $O/prof_$n; }
prof c2 --steps 60 --warmup 10 --in-flight 1 --no-cpu-baseline --sustained-seconds 0 --no-fp16-leg
prof c2_two --steps 60 --warmup 10 --no-cpu-baseline --sustained-seconds 0 --no-fp16-leg
head -30 $O/r06p_c2_graph_timeline.txt | cut -c1-130
