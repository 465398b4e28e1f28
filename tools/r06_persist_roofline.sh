#!/bin/bash
# round 6: the fused conv3d_2b -> 2c -> pool call's duration as bench.py's roofline measures it (prefix-graph differences), persistent tile loop on / off, same box, interleaved
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd $R
for rep in 1 2; do for v in 1 0; do
  timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-fp16-leg --sustained-seconds 0 --opt conv_persist=$v 2>/dev/null | grep '^{' > $O/persist_rl_${v}_$rep.json
  python - <<P
import json
j = json.load(open("$O/persist_rl_${v}_$rep.json")); r = j["roofline"]
kt = {}
for row in j["kernel_table"]:
    kt.setdefault(row["kernel"], []).append(row["us"])
c = [k for k in kt if "pre_pool" in k][0]
print("conv_persist=$v rep $rep: value %.0f one-batch %.4f ms kernel-time %.4f | %s %.1f us + seam %.1f" % (j["value"], j["one_batch_in_flight"]["ms_per_step"], j["kernel_time_ms_per_step"], c[:50], kt[c][0], kt.get("pool_seam_fix_kernel", [0])[0]))
P
done; done
