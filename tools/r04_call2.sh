#!/bin/bash
# round 4, call 2: GPU test suite at the new ABI (27), C2 with / without the stem + pool fusion, C4 at 8 clips with the roi list
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -15 > $O/r04b_gputests.log; tail -5 $O/r04b_gputests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
for i in 1 2; do
timeout 300 python bench.py --no-cpu-baseline --sustained-seconds 0 > $O/r04b_c2_fused_$i.json 2> $O/r04b_c2_fused_$i.err
timeout 300 python tools/bench_with.py backbone.FUSE_STEM_POOL=False -- --no-cpu-baseline --sustained-seconds 0 > $O/r04b_c2_unfused_$i.json 2> $O/r04b_c2_unfused_$i.err
done
timeout 300 python bench.py --config c3 --no-cpu-baseline --steps 30 --warmup 5 > $O/r04b_c3.json 2> $O/r04b_c3.err
timeout 300 python bench.py --config c5 --no-cpu-baseline > $O/r04b_c5.json 2> $O/r04b_c5.err
timeout 500 python bench.py --config c4 --dtype bf16 --clips 8 --tubes 15 --steps 10 --warmup 3 --no-cpu-baseline > $O/r04b_c4_bf16_b8_t15.json 2> $O/r04b_c4_bf16_b8_t15.err
timeout 500 python bench.py --config c4 --dtype bf16 --steps 10 --warmup 3 --no-cpu-baseline > $O/r04b_c4_bf16_b1_t5.json 2> $O/r04b_c4_bf16_b1_t5.err
python - <<P
import json,glob
for f in sorted(glob.glob('$O/r04b_*.json')):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1])
        r=j.get('roofline',{})
        print(f.split('/')[-1], j['value'], j['ms_per_step'], (j.get('one_batch_in_flight') or {}).get('value'), r.get('kernel','')[:60], r.get('frac'), r.get('avg_launch_ms'), [ (k['kernel'][11:40],k['avg_launch_ms']) for k in r.get('next_kernels',[])])
    except Exception as e:
        print(f, 'ERR', e)
P
tail -3 $O/r04b_*.err | tail -30
