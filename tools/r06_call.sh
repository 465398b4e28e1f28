#!/bin/bash
# one gpurun call of round 6: what the driver runs at round end (GPU tests, smoke, its exact bench command) + whatever A/B the call names
#   tools/r06_call.sh <tag> [tests|notests] [extra command ...]
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
TAG=${1:-x}; shift
T=${1:-tests}; shift
mkdir -p $O
cd $R
if [ "$T" = tests ]; then
  timeout 1700 python -m pytest tests -x -q -m gpu > $O/gputests_$TAG.log 2>&1; grep -E "passed|failed|error" $O/gputests_$TAG.log | tail -3
  timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
fi
# the driver's exact command (tabulated per lease in profiles/r06_driver_command.txt)
timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/driver_cmd_$TAG.json 2> $O/driver_cmd_$TAG.err
python - <<PY
import json
try:
    d = json.load(open("$O/driver_cmd_$TAG.json"))
    print("driver_cmd", d["value"], d["ms_per_step"], "one", d["one_batch_in_flight"]["ms_per_step"], "kt", d["kernel_time_ms_per_step"], "fp16", (d.get("fp16") or {}).get("value"),
          "clk", d.get("clock_ghz", {}).get("one_batch_loop"))
    r = d["roofline"]; print(r["kernel"][:60], r["frac"], r["avg_launch_ms"])
except Exception as e:
    print("driver_cmd failed", e); print(open("$O/driver_cmd_$TAG.err").read()[-1500:])
PY
for c in "$@"; do
  echo "== $c"
  eval "timeout 900 $c" 2>&1 | tail -25
done
