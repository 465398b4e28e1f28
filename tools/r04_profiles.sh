#!/bin/bash
# round 4: rocprofv3 kernel statistics of the bench configs, the replayed-step timeline, and the two PMC traffic passes for C2 and C4
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
export STEP_COMMIT=${STEP_COMMIT:-?}
prof() { n=$1; shift
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$n -- python $R/bench.py "$@" > $O/r04p_${n}_prof.json 2> $O/r04p_${n}_prof.err
  python $R/tools/prof_summary.py $O/prof_$n $O/r04p_${n}_kernel_stats.txt > /dev/null 2>&1
  if [ "$n" = "c2" ]; then python $R/tools/graph_timeline.py $O/prof_$n > $O/r04p_c2_graph_timeline.txt 2>&1; fi
  rm -rf $O/prof_$n; }
prof c2 --steps 60 --warmup 10 --in-flight 1 --no-cpu-baseline --sustained-seconds 0
prof c2_two --steps 60 --warmup 10 --no-cpu-baseline --sustained-seconds 0
prof c5 --config c5 --steps 20 --warmup 5 --no-cpu-baseline --sustained-seconds 0
prof c3 --config c3 --steps 20 --warmup 5 --no-cpu-baseline --in-flight 1
prof c4_bf16 --config c4 --dtype bf16 --steps 10 --warmup 3 --no-cpu-baseline --no-graph
prof c4_bf16_b8_t15 --config c4 --dtype bf16 --clips 8 --tubes 15 --steps 6 --warmup 2 --no-cpu-baseline --no-graph
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$c -- python $R/bench.py --steps 4 --warmup 2 --no-graph --no-cpu-baseline --sustained-seconds 0 > $O/pmc_$c.json 2> $O/pmc_$c.err
  timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc4_$c -- python $R/bench.py --config c4 --dtype bf16 --clips 8 --tubes 15 --steps 2 --warmup 1 --no-graph --no-cpu-baseline > $O/pmc4_$c.json 2> $O/pmc4_$c.err
  timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc41_$c -- python $R/bench.py --config c4 --dtype bf16 --steps 3 --warmup 1 --no-graph --no-cpu-baseline > $O/pmc41_$c.json 2> $O/pmc41_$c.err
done
cd $R
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
subprocess.call([sys.executable, "tools/pmc_traffic.py", "$O/pmc4_FETCH_SIZE", "$O/pmc4_WRITE_SIZE", "$O/traffic_c4.json"] + names("$O/pmc4_FETCH_SIZE"), stdout=open("$O/pmc_traffic_c4.log", "w"), stderr=subprocess.STDOUT)
j = json.load(open("$O/traffic_c2.json")); k = j["kernels"]
for K1, K11 in (('void step::conv_tap_kernel<step::bf16_t, 3, 3, 3, 3, 3, 2, 2, 8, 1>(step::ConvParams)', 'void step::conv_tap_kernel<step::bf16_t, 3, 1, 3, 3, 3, 2, 2, 8, 1>(step::ConvParams)'),
                ('void step::conv_tap_pre_kernel<step::bf16_t, 3, 3>(step::ConvParams)', 'void step::conv_tap_pre_kernel<step::bf16_t, 3, 1>(step::ConvParams)')):
    if K11 in k: k[K11]['with'] = K1
try:
    j4 = json.load(open("$O/traffic_c4.json"))
    j["kernels_c4"] = j4["kernels"]; j["commit_c4"] = j4.get("commit")
    j["note_c4"] = "C4 training step, bf16, 8 clips x 15 tubes per GPU, eager (bench.py --config c4 --dtype bf16 --clips 8 --tubes 15 --no-graph)"
    j["workload_c4"] = {"clips": 8, "tubes": 15}
    subprocess.call([sys.executable, "tools/pmc_traffic.py", "$O/pmc41_FETCH_SIZE", "$O/pmc41_WRITE_SIZE", "$O/traffic_c4_b1.json"] + names("$O/pmc41_FETCH_SIZE"), stdout=open("$O/pmc_traffic_c4_b1.log", "w"), stderr=subprocess.STDOUT)
    j41 = json.load(open("$O/traffic_c4_b1.json"))
    j["kernels_c4_b1"] = j41["kernels"]; j["commit_c4_b1"] = j41.get("commit")
    j["note_c4_b1"] = "C4 training step, bf16, 1 clip x 5 tubes per GPU, eager (bench.py --config c4 --dtype bf16 --no-graph)"
    j["workload_c4_b1"] = {"clips": 1, "tubes": 5}
except Exception as e:
    print("no c4 traffic:", e)
json.dump(j, open("$O/traffic_latest.json", "w"), indent=1)
for n_, v in sorted(j["kernels"].items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"])[:6]:
    print("c2", n_[:80], v["hbm_bytes_per_launch"], v["read_bytes"], v["write_bytes"])
for n_, v in sorted(j.get("kernels_c4", {}).items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"])[:8]:
    print("c4", n_[:80], v["hbm_bytes_per_launch"], v["read_bytes"], v["write_bytes"], v["launches_sampled"])
P
rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmc4_FETCH_SIZE $O/pmc4_WRITE_SIZE $O/pmc41_FETCH_SIZE $O/pmc41_WRITE_SIZE
head -34 $O/r04p_c2_graph_timeline.txt | cut -c1-150; head -12 $O/r04p_c2_kernel_stats.txt | cut -c1-150
