#!/bin/bash
# round 6: the bench lines of every config on one box (no profiler), one JSON line each
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd $R
run() { n=$1; shift; timeout 400 python bench.py "$@" 2>$O/r06b_$n.err | grep '^{' > $O/r06b_$n.json; }
run c2 --feed u8
run c2_one --in-flight 1 --no-cpu-baseline
run c5 --config c5 --no-cpu-baseline
run c3 --config c3 --steps 30 --warmup 5
run c3_34 --config c3 --tubes 34 --steps 20 --warmup 5 --no-cpu-baseline
run c4_bf16 --config c4 --dtype bf16 --steps 30 --warmup 3 --no-cpu-baseline
run c4_bf16_x --config c4 --dtype bf16 --steps 30 --warmup 3 --no-cpu-baseline --force-exchange
run c4_bf16_b8_t15 --config c4 --dtype bf16 --clips 8 --tubes 15 --steps 10 --warmup 3 --no-cpu-baseline
run c4_f32 --config c4 --steps 10 --warmup 3 --no-cpu-baseline
run c4_select --config c4 --dtype bf16 --steps 30 --warmup 3 --no-cpu-baseline --select
run c4_select_eager --config c4 --dtype bf16 --steps 10 --warmup 3 --no-cpu-baseline --select --no-graph
run c4_select_b8 --config c4 --dtype bf16 --clips 8 --steps 10 --warmup 3 --no-cpu-baseline --select
run c4_bf16_t15 --config c4 --dtype bf16 --tubes 15 --steps 30 --warmup 3 --no-cpu-baseline
# the torchrun entry, fed from pinned host uint8 frames against resident clips, one rank
for f in none u8; do
  timeout 300 python train_step_amd.py --iters 40 --warmup-iters 3 --log-every 0 --feed $f 2>$O/r06b_train_$f.err | grep summary > $O/r06b_train_$f.json
  timeout 300 python train_step_amd.py --iters 40 --warmup-iters 3 --log-every 0 --feed $f --select 2>$O/r06b_train_sel_$f.err | grep summary > $O/r06b_train_sel_$f.json
done
python - <<P
import json
for n in ("train_none", "train_u8", "train_sel_none", "train_sel_u8"):
    try:
        j=json.load(open("$O/r06b_%s.json"%n)); print("%-16s train_step_amd.py: %.3f ms per iteration, %.2f clips/s, %s, feed %s" % (n, j["ms_per_iter"], j["clips_per_s"], j["launch"], j["feed"]))
    except Exception as e:
        print(n, "ERR", e)
for n in ("c2","c2_one","c5","c3","c3_34","c4_bf16","c4_bf16_x","c4_bf16_b8_t15","c4_f32","c4_select","c4_select_eager","c4_select_b8","c4_bf16_t15"):
    try:
        j=json.load(open("$O/r06b_%s.json"%n)); r=j.get("roofline",{})
        print("%-16s %9.2f %s  ms/step %.4f  one %s  sus %s  | %s frac %s | fed %s" % (n, j["value"], j["unit"], j["ms_per_step"], j.get("one_batch_in_flight",{}).get("value"),
              j.get("sustained",{}).get("value"), r.get("kernel","")[:44], r.get("frac"), (j.get("fed") or {}).get("value")))
    except Exception as e:
        print(n, "ERR", e)
P
