#!/bin/bash
# round 4, call 3: the restructured training step (block-level autograd nodes, fused head tail, grouped weight-gradient sums)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -15 > $O/r04c_gputests.log; tail -4 $O/r04c_gputests.log
timeout 500 python bench.py --config c4 --dtype bf16 --steps 10 --warmup 3 --no-cpu-baseline > $O/r04c_c4_bf16_b1_t5.json 2> $O/r04c_c4_bf16_b1_t5.err
timeout 500 python bench.py --config c4 --dtype bf16 --clips 8 --tubes 15 --steps 10 --warmup 3 --no-cpu-baseline > $O/r04c_c4_bf16_b8_t15.json 2> $O/r04c_c4_bf16_b8_t15.err
timeout 500 python bench.py --config c4 --steps 10 --warmup 3 --no-cpu-baseline > $O/r04c_c4_f32_b1_t5.json 2> $O/r04c_c4_f32_b1_t5.err
timeout 300 python bench.py --config c3 --no-cpu-baseline --steps 30 --warmup 5 > $O/r04c_c3.json 2> $O/r04c_c3.err
timeout 300 python bench.py --no-cpu-baseline > $O/r04c_c2.json 2> $O/r04c_c2.err
cd /tmp; export TMPDIR=/tmp
prof() { n=$1; shift
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$n -- python $R/bench.py "$@" > $O/r04c_${n}_prof.json 2> $O/r04c_${n}_prof.err
  python $R/tools/prof_summary.py $O/prof_$n $O/r04c_${n}_kernel_stats.txt > /dev/null 2>&1
  rm -rf $O/prof_$n; }
prof c4_bf16 --config c4 --dtype bf16 --steps 10 --warmup 3 --no-cpu-baseline
prof c4_bf16_b8 --config c4 --dtype bf16 --clips 8 --tubes 15 --steps 6 --warmup 2 --no-cpu-baseline
prof c3 --config c3 --steps 20 --warmup 5 --no-cpu-baseline --in-flight 1
cd $R
python - <<P
import json,glob
for f in sorted(glob.glob('$O/r04c_*.json')):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1])
        r=j.get('roofline',{})
        print(f.split('/')[-1], j['value'], j['ms_per_step'], (j.get('one_batch_in_flight') or {}).get('value'), (j.get('sustained') or {}), r.get('kernel','')[:60], r.get('frac'))
    except Exception as e:
        print(f, 'ERR', e)
P
for f in $O/r04c_*.err; do echo "== $f"; tail -n 3 $f; done 2>/dev/null | tail -40
head -8 $O/r04c_c4_bf16_kernel_stats.txt | cut -c1-160
