#!/bin/bash
# round 6: the captured selected training step -- its test, then the bench lines of the fixed-tube step, the selected step captured and eager
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd $R
timeout 900 python -m pytest tests/test_gpu_graph_step.py -x -q -k "selected" 2>&1 | tail -30
run() { n=$1; shift; timeout 600 python bench.py --config c4 --dtype bf16 --no-cpu-baseline "$@" 2>$O/c4sel_$n.err | grep '^{' > $O/c4sel_$n.json
  python - <<P
import json
try:
    j = json.load(open("$O/c4sel_$n.json")); print("$n", j["value"], j["ms_per_step"], j["config"]["launch"][:100])
except Exception as e:
    print("$n ERR", e); print(open("$O/c4sel_$n.err").read()[-1500:])
P
}
run fixed --steps 30 --warmup 3
run select --steps 30 --warmup 3 --select
run select_eager --steps 10 --warmup 3 --select --no-graph
run select_b8 --steps 10 --warmup 3 --select --clips 8
run fixed_b8_t15 --steps 10 --warmup 3 --clips 8 --tubes 15
