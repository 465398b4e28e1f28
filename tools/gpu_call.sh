#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd $R
timeout 600 python bench.py --steps 50 --warmup 10 --no-cpu-baseline > $O/bench_c2.json 2> $O/bench_c2.err; tail -3 $O/bench_c2.err; python -c "
import json; d=json.load(open('$O/bench_c2.json')); print(d['value'], d['ms_per_step']); r=d['roofline']; print(r['kernel'][:60], r['frac'], r['avg_launch_ms'], r['traffic']); 
for k in r['next_kernels']: print(k['kernel'][:70], k['frac'], k['avg_launch_ms'], k['launches_per_step'], k['traffic'])"
