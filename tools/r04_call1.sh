#!/bin/bash
# round 4, call 1: baselines at the round's start with the new bench fields (sustained loop, c3 / c4 cpu_baseline, --clips / --tubes)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 400 python bench.py > $O/r04a_c2.json 2> $O/r04a_c2.err; tail -c 600 $O/r04a_c2.err
for cfg in "1 5" "1 15" "8 5" "8 15"; do set -- $cfg
  timeout 500 python bench.py --config c4 --dtype bf16 --clips $1 --tubes $2 --steps 10 --warmup 3 --no-cpu-baseline > $O/r04a_c4_bf16_b$1_t$2.json 2> $O/r04a_c4_bf16_b$1_t$2.err
  tail -c 300 $O/r04a_c4_bf16_b$1_t$2.err
done
timeout 500 python bench.py --config c4 --clips 8 --tubes 15 --steps 5 --warmup 2 --no-cpu-baseline > $O/r04a_c4_f32_b8_t15.json 2> $O/r04a_c4_f32_b8_t15.err
timeout 500 python bench.py --config c4 --dtype bf16 --steps 10 --warmup 3 > $O/r04a_c4_bf16_cpu.json 2> $O/r04a_c4_bf16_cpu.err
timeout 500 python bench.py --config c3 --steps 20 --warmup 5 > $O/r04a_c3.json 2> $O/r04a_c3.err
timeout 600 python tools/cpu_baselines.py --budget 15 > $O/r04a_cpu_baselines.txt 2> $O/r04a_cpu_baselines.err
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c4b8 -- python $R/bench.py --config c4 --dtype bf16 --clips 8 --tubes 15 --steps 6 --warmup 2 --no-cpu-baseline > $O/r04a_c4_b8_prof.json 2> $O/r04a_c4_b8_prof.err
python $R/tools/prof_summary.py $O/prof_c4b8 $O/r04a_c4_bf16_b8_t15_kernel_stats.txt > /dev/null 2>&1
rm -rf $O/prof_c4b8
cd $R
python - <<P
import json,glob
for f in sorted(glob.glob('$O/r04a_*.json')):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], j['value'], j['ms_per_step'], j.get('sustained',{}).get('value'), j.get('sustained',{}).get('clock_ghz'), (j.get('cpu_baseline') or {}).get('value'), j.get('roofline',{}).get('kernel','')[:60], j.get('roofline',{}).get('frac'))
    except Exception as e:
        print(f, 'ERR', e)
P
cat $O/r04a_cpu_baselines.txt
