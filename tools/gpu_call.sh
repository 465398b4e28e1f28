#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -4 > $O/gputests.log; cat $O/gputests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; cut -c1-120 $O/bench_default.json
