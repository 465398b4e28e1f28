#!/bin/bash
# round 6: rocprofv3 kernel statistics of the bench configs, the replayed-step timeline, SQ counters of the C2 kernels and the PMC traffic
# passes for C2 and C3 (separate --pmc passes, --kernel-trace only beside them)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
export STEP_COMMIT=${STEP_COMMIT:-?}
prof() { n=$1; shift
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$n -- python $R/bench.py "$@" 2> $O/r06p_${n}_prof.err | grep '^{' > $O/r06p_${n}_prof.json
  python $R/tools/prof_summary.py $O/prof_$n $O/r06p_${n}_kernel_stats.txt > /dev/null 2>&1
  if [ "$n" = "c2" ]; then python $R/tools/graph_timeline.py $O/prof_$n > $O/r06p_c2_graph_timeline.txt 2>&1; fi
  rm -rf $O/prof_$n; }
prof c2 --steps 60 --warmup 10 --in-flight 1 --no-cpu-baseline --sustained-seconds 0 --no-fp16-leg
prof c2_two --steps 60 --warmup 10 --no-cpu-baseline --sustained-seconds 0 --no-fp16-leg
prof c5 --config c5 --steps 20 --warmup 5 --no-cpu-baseline --sustained-seconds 0
prof c3 --config c3 --steps 20 --warmup 5 --no-cpu-baseline --in-flight 1
prof c3_34 --config c3 --tubes 34 --steps 20 --warmup 5 --no-cpu-baseline --in-flight 1
prof c4_bf16 --config c4 --dtype bf16 --steps 10 --warmup 3 --no-cpu-baseline --no-graph
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$c -- python $R/bench.py --steps 4 --warmup 2 --no-graph --no-cpu-baseline --sustained-seconds 0 > /dev/null 2> $O/pmc_$c.err
  timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc3_$c -- python $R/bench.py --config c3 --steps 3 --warmup 1 --no-graph --no-cpu-baseline --in-flight 1 > /dev/null 2> $O/pmc3_$c.err
  timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc334_$c -- python $R/bench.py --config c3 --tubes 34 --steps 3 --warmup 1 --no-graph --no-cpu-baseline --in-flight 1 > /dev/null 2> $O/pmc334_$c.err
  timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc4_$c -- python $R/bench.py --config c4 --dtype bf16 --clips 8 --tubes 15 --steps 2 --warmup 1 --no-graph --no-cpu-baseline > /dev/null 2> $O/pmc4_$c.err
  timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc41_$c -- python $R/bench.py --config c4 --dtype bf16 --steps 3 --warmup 1 --no-graph --no-cpu-baseline > /dev/null 2> $O/pmc41_$c.err
done
i=0
for grp in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD"; do
  i=$((i+1))
  timeout 400 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $O/sq2_$i -- python $R/bench.py --steps 4 --warmup 2 --no-graph --no-cpu-baseline --sustained-seconds 0 > /dev/null 2> $O/sq2_$i.err
done
cd $R
{ echo "# rocprofv3 --pmc (three separate passes) of bench.py --steps 4 --warmup 2 --no-graph (C2, bf16, 8 clips): per-launch means; MFMA busy share = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x SQ_BUSY_CU_CYCLES)"
  for k in stem_stream_kernel conv_tap_pre_pool_persist_kernel conv_tap_pre_pool_kernel pool_seam_fix_kernel conv_tap_group_pw_kernel conv_tap_group_kernel pool333_pw_kernel; do python tools/pmc_dump.py $k $O/sq2_1 $O/sq2_2 $O/sq2_3; done; } > $O/r06_pmc_c2.txt
python - <<P
import csv, glob, subprocess, sys, json
def names(d):
    s = set()
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Kernel_Name"].startswith("void step::") or r["Kernel_Name"].startswith("step::"):
                s.add(r["Kernel_Name"])
    return sorted(s)
subprocess.call([sys.executable, "tools/pmc_traffic.py", "$O/pmc_FETCH_SIZE", "$O/pmc_WRITE_SIZE", "$O/traffic_c2.json"] + names("$O/pmc_FETCH_SIZE"), stdout=open("$O/pmc_traffic_c2.log", "w"), stderr=subprocess.STDOUT)
subprocess.call([sys.executable, "tools/pmc_traffic.py", "$O/pmc3_FETCH_SIZE", "$O/pmc3_WRITE_SIZE", "$O/traffic_c3.json"] + names("$O/pmc3_FETCH_SIZE"), stdout=open("$O/pmc_traffic_c3.log", "w"), stderr=subprocess.STDOUT)
for tag in ("334", "4", "41"):
    subprocess.call([sys.executable, "tools/pmc_traffic.py", "$O/pmc%s_FETCH_SIZE" % tag, "$O/pmc%s_WRITE_SIZE" % tag, "$O/traffic_c%s.json" % tag] + names("$O/pmc%s_FETCH_SIZE" % tag),
                    stdout=open("$O/pmc_traffic_c%s.log" % tag, "w"), stderr=subprocess.STDOUT)
j = json.load(open("$O/traffic_c2.json")); k = j["kernels"]
for K1, K11 in (('void step::conv_tap_pre_pool_persist_kernel<step::bf16_t, 3>(step::ConvParams)', 'void step::conv_tap_pre_pool_kernel<step::bf16_t, 1>(step::ConvParams)'),
                ('void step::conv_tap_pre_pool_kernel<step::bf16_t, 3>(step::ConvParams)', 'void step::conv_tap_pre_pool_kernel<step::bf16_t, 1>(step::ConvParams)'),
                ('void step::conv_tap_pre_kernel<step::bf16_t, 3, 3>(step::ConvParams)', 'void step::conv_tap_pre_kernel<step::bf16_t, 3, 1>(step::ConvParams)')):
    if K11 in k: k[K11]['with'] = K1
for tag, key, note, wl in (("4", "c4", "C4 training step, bf16, 8 clips x 15 tubes per GPU, eager (bench.py --config c4 --dtype bf16 --clips 8 --tubes 15 --no-graph)", {"clips": 8, "tubes": 15}),
                           ("41", "c4_b1", "C4 training step, bf16, 1 clip x 5 tubes per GPU, eager (bench.py --config c4 --dtype bf16 --no-graph)", {"clips": 1, "tubes": 5}),
                           ("334", "c3_34", "C3 full inference, bf16, 4 clips x 34 tubes per GPU (the reference's default), eager (bench.py --config c3 --tubes 34 --no-graph --in-flight 1)", {"clips": 4, "tubes": 34})):
    try:
        jj = json.load(open("$O/traffic_c%s.json" % tag))
        j["kernels_" + key] = jj["kernels"]; j["commit_" + key] = jj.get("commit"); j["note_" + key] = note; j["workload_" + key] = wl
    except Exception as e:
        print("no %s traffic:" % key, e)
try:
    j3 = json.load(open("$O/traffic_c3.json"))
    j["kernels_c3"] = j3["kernels"]; j["commit_c3"] = j3.get("commit")
    j["note_c3"] = "C3 full inference, bf16, 4 clips x 11 tubes per GPU, eager (bench.py --config c3 --no-graph --in-flight 1)"
    j["workload_c3"] = {"clips": 4, "tubes": 11}
except Exception as e:
    print("no c3 traffic:", e)
json.dump(j, open("$O/traffic_latest.json", "w"), indent=1)
for n_, v in sorted(j["kernels"].items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"])[:6]:
    print("c2", n_[:80], v["hbm_bytes_per_launch"], v["read_bytes"], v["write_bytes"])
for n_, v in sorted(j.get("kernels_c3", {}).items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"])[:6]:
    print("c3", n_[:80], v["hbm_bytes_per_launch"], v["read_bytes"], v["write_bytes"], v["launches_sampled"])
P
rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmc3_FETCH_SIZE $O/pmc3_WRITE_SIZE $O/pmc334_* $O/pmc4_* $O/pmc41_* $O/sq2_1 $O/sq2_2 $O/sq2_3
head -34 $O/r06p_c2_graph_timeline.txt | cut -c1-150; head -12 $O/r06p_c2_kernel_stats.txt | cut -c1-150
for f in c2 c2_two c5 c3 c3_34 c4_bf16; do python -c "
import json,sys
try:
    j=json.load(open('$O/r06p_${f}_prof.json')); print('$f', j['value'], j['ms_per_step'], j.get('one_batch_in_flight',{}).get('value'), j['roofline']['kernel'][:50], j['roofline']['frac'])
except Exception as e: print('$f', 'ERR', e)
"; done
