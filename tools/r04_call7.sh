#!/bin/bash
# round 4, GPU call 7: wgrad A/B (wide-halo form), the kernel / graph / ddp GPU tests, every config's bench line
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd $R
[ -f tools/libstep_amd_prev.so ] && timeout 200 python tools/wgrad_bench.py --dtype bf16 --libs prev --iters 10 --only 2c@,3c_b1b@400x8,4f_b1b 2>&1 | tail -8 | tee $O/r04e_wgrad_ab.txt
timeout 900 python -m pytest tests -x -q -m gpu -k "kernel or ddp or graph" 2>&1 | grep -E "passed|failed"
for cfg in "1 5" "1 15" "8 5" "8 15"; do set -- $cfg
  timeout 500 python bench.py --config c4 --dtype bf16 --clips $1 --tubes $2 --steps 10 --warmup 3 --no-cpu-baseline > $O/r04e_c4_bf16_b$1_t$2.json 2> $O/r04e_c4_bf16_b$1_t$2.err
done
timeout 300 python bench.py --config c3 --no-cpu-baseline --steps 30 --warmup 5 > $O/r04e_c3.json 2> $O/r04e_c3.err
timeout 300 python bench.py --config c5 --no-cpu-baseline > $O/r04e_c5.json 2> $O/r04e_c5.err
timeout 300 python bench.py --config c2 --no-cpu-baseline > $O/r04e_c2.json 2> $O/r04e_c2.err
python - <<P
import json,glob
for f in sorted(glob.glob('$O/r04e_*.json')):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1]); r=j.get('roofline',{})
        print(f.split('/')[-1], j['value'], j['ms_per_step'], (j.get('one_batch_in_flight') or {}).get('value'), (j.get('sustained') or {}).get('value'), r.get('kernel','')[:60], r.get('frac'), r.get('traffic'), j.get('launches_per_step'))
    except Exception as e:
        print(f, 'ERR', e)
P
