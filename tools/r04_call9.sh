#!/bin/bash
# round 4, GPU call 9: weight-gradient A/B (pixel-stream pointwise form, parallel fixed-order sums) and the training-step configs
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd $R
[ -f tools/libstep_amd_prev.so ] && timeout 300 python tools/wgrad_bench.py --graph --dtype bf16 --libs prev --iters 10 2>&1 | tail -32 | tee $O/r04g_wgrad_ab.txt
timeout 900 python -m pytest tests -x -q -m gpu -k "kernel or ddp or graph" 2>&1 | grep -E "passed|failed|Error|error" | head -5
for cfg in "1 5" "8 15"; do set -- $cfg
  timeout 500 python bench.py --config c4 --dtype bf16 --clips $1 --tubes $2 --steps 10 --warmup 3 --no-cpu-baseline > $O/r04g_c4_bf16_b$1_t$2.json 2> $O/r04g_c4_bf16_b$1_t$2.err
done
timeout 500 python bench.py --config c4 --steps 10 --warmup 3 --no-cpu-baseline > $O/r04g_c4_f32_b1_t5.json 2> $O/r04g_c4_f32_b1_t5.err
python - <<P
import json,glob
for f in sorted(glob.glob('$O/r04g_*.json')):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1]); r=j.get('roofline',{})
        print(f.split('/')[-1], j['value'], j['ms_per_step'], r.get('kernel','')[:60], r.get('frac'), r.get('launches_per_step'))
    except Exception as e:
        print(f, 'ERR', e)
P
tail -n 3 $O/r04g_*.err | tail -20
