#!/bin/bash
# round 4, final GPU calls: every config's bench line (profiles/r04_bench_all_configs.txt), then the GPU suite + smoke + default bench
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd $R
rm -f $O/r04h_*.json
for cfg in "1 5" "1 15" "8 5" "8 15"; do set -- $cfg
  nocpu="--no-cpu-baseline"; [ "$1 $2" = "1 5" ] && nocpu=""          # (the CPU leg is one clip either way: timed once)
  timeout 500 python bench.py --config c4 --dtype bf16 --clips $1 --tubes $2 --steps 10 --warmup 3 $nocpu > $O/r04h_c4_bf16_b$1_t$2.json 2> $O/r04h_c4_bf16_b$1_t$2.err
done
timeout 500 python bench.py --config c4 --steps 10 --warmup 3 --no-cpu-baseline > $O/r04h_c4_f32_b1_t5.json 2> $O/r04h_c4_f32_b1_t5.err
timeout 300 python bench.py --config c3 --steps 30 --warmup 5 > $O/r04h_c3.json 2> $O/r04h_c3.err
timeout 300 python bench.py --config c5 --no-cpu-baseline > $O/r04h_c5.json 2> $O/r04h_c5.err
timeout 300 python bench.py --config c2 > $O/r04h_c2.json 2> $O/r04h_c2.err
python - <<P > $O/r04_bench_all_configs.txt
import json,glob
print("# bench.py lines of one gpurun call (one MI355X, tools/r04_call10.sh): value | ms_per_step | one batch in flight | sustained | dominant kernel | roofline frac | PMC traffic per launch | cpu_baseline")
for f in sorted(glob.glob('$O/r04h_*.json')):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1]); r=j.get('roofline',{})
        print(f.split('/')[-1], '|', j['value'], j['unit'], '|', j['ms_per_step'], 'ms |', (j.get('one_batch_in_flight') or {}).get('value'), '|', (j.get('sustained') or {}).get('value'), '|',
              r.get('kernel','')[:70], '|', r.get('frac'), '|', r.get('traffic'), '|', (j.get('cpu_baseline') or {}).get('value'), (j.get('cpu_baseline') or {}).get('unit'), (j.get('cpu_baseline') or {}).get('cores'))
    except Exception as e:
        print(f, 'ERR', e)
print()
for f in sorted(glob.glob('$O/r04h_*.json')):
    print('##', f.split('/')[-1]); print(open(f).read().strip().splitlines()[-1]); print()
P
head -12 $O/r04_bench_all_configs.txt | cut -c1-260
bash tools/gpu_call.sh
