#!/bin/bash
# round 6: C3 full inference at 11 and at 34 tubes per clip (the reference's default), bench lines + the kernel statistics of the 34-tube run
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd $R
for t in 11 34; do
  timeout 400 python bench.py --config c3 --tubes $t --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | grep '^{' > $O/r06_c3_t$t.json
  python -c "
import json; j=json.load(open('$O/r06_c3_t$t.json')); print('c3 tubes $t:', j['value'], 'clips/s', j['ms_per_step'], 'ms; one at a time', j['one_batch_in_flight']['value'], '|', j['roofline']['kernel'][:50], j['roofline']['frac'])"
done
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c3_34 -- python $R/bench.py --config c3 --tubes 34 --steps 20 --warmup 5 --no-cpu-baseline --in-flight 1 > /dev/null 2>&1
python $R/tools/prof_summary.py $O/prof_c3_34 $O/r06p_c3_34_kernel_stats.txt > /dev/null 2>&1; rm -rf $O/prof_c3_34
head -12 $O/r06p_c3_34_kernel_stats.txt | cut -c1-150
