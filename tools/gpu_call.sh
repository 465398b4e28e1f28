#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 600 python bench.py --verbose > $O/bench_default.json 2> $O/bench_default.err
cat $O/bench_default.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline'], d['kernel_time_ms_per_step'])"; grep "ms " $O/bench_default.err | head -20
