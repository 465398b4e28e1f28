#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 1200 python -m pytest tests -q -m gpu 2>&1 | tail -6 > $O/gputests.log; cat $O/gputests.log
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; cut -c1-330 $O/bench_default.json
python -c "
import json; d=json.load(open('$O/bench_default.json')); print(d['roofline']); print(d['cpu_baseline'])"
