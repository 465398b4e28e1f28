#!/bin/bash
# one gpurun call: the GPU test suite, smoke() and the default bench line (what the driver runs at round end)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -x -q -m gpu > $O/gputests_full.log 2>&1; grep -E "passed|failed|error" $O/gputests_full.log | tail -5 > $O/gputests.log; cat $O/gputests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; python -c "
import json; d=json.load(open('$O/bench_default.json')); print(d['value'], d['ms_per_step']); r=d['roofline']; print(r['kernel'][:70], r['frac'], r['avg_launch_ms'], r['traffic']); print([(k['kernel'][11:45], k['frac']) for k in r['next_kernels']])"
