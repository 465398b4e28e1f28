#!/bin/bash
# one gpurun call: rocprofv3 kernel statistics of the bench configs and the two PMC traffic passes (outputs under gpurun_out/,
# the summaries are then copied to profiles/ by hand)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
prof() { n=$1; shift
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$n -- python $R/bench.py "$@" > $O/bench_${n}_prof.json 2> $O/bench_${n}_prof.err
  python $R/tools/prof_summary.py $O/prof_$n $O/prof_${n}_summary.txt > /dev/null 2>&1
  cp $(find $O/prof_$n -name "*kernel_stats.csv" | head -1) $O/prof_${n}_kernel_stats.csv 2>/dev/null
  rm -rf $O/prof_$n; }
# per-kernel durations are defined one batch at a time (what bench.py's roofline measures: prefix replays of ONE captured step);
# the default command keeps two batches in flight, where launches of the two batches share the CUs and a trace's per-launch
# durations are no longer per-kernel costs -- both traces are kept
prof c2 --steps 100 --warmup 10 --in-flight 1 --no-cpu-baseline
prof c2_two --steps 100 --warmup 10
prof c3 --config c3 --steps 20 --warmup 5 --no-cpu-baseline
prof c4_bf16 --config c4 --dtype bf16 --steps 10 --warmup 3 --no-cpu-baseline
prof c4_f32 --config c4 --steps 10 --warmup 3 --no-cpu-baseline
prof c5 --config c5 --steps 20 --warmup 5 --no-cpu-baseline
export STEP_COMMIT=${STEP_COMMIT:-?}
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$c -- python $R/bench.py --steps 4 --warmup 2 --no-graph --no-cpu-baseline > $O/pmc_$c.json 2> $O/pmc_$c.err
done
cd $R
# every step:: kernel of the C2 backbone that the PMC passes saw (names as rocprofv3 prints them)
python - <<P
import csv, glob, subprocess, sys
names = set()
for f in glob.glob('$O/pmc_FETCH_SIZE/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Kernel_Name"].startswith("void step::"):
            names.add(r["Kernel_Name"])
sys.exit(subprocess.call([sys.executable, "tools/pmc_traffic.py", "$O/pmc_FETCH_SIZE", "$O/pmc_WRITE_SIZE", "$O/traffic_latest.json"] + sorted(names),
                         stdout=open("$O/pmc_traffic.log", "w"), stderr=subprocess.STDOUT))
P
python - <<P
import json
f='$O/traffic_latest.json'; j=json.load(open(f)); k=j['kernels']
# conv3d_2c's partial last round (C2) is a second launch of the same call: plain form and the form with conv3d_2b fused in
for K1, K11 in (('void step::conv_tap_kernel<step::bf16_t, 3, 3, 3, 3, 3, 2, 2, 8, 1>(step::ConvParams)', 'void step::conv_tap_kernel<step::bf16_t, 3, 1, 3, 3, 3, 2, 2, 8, 1>(step::ConvParams)'),
                ('void step::conv_tap_pre_kernel<step::bf16_t, 3, 3>(step::ConvParams)', 'void step::conv_tap_pre_kernel<step::bf16_t, 3, 1>(step::ConvParams)')):
    if K11 in k: k[K11]['with']=K1
json.dump(j, open(f,'w'), indent=1)
P
rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE
head -22 $O/prof_c2_summary.txt | cut -c1-200; cat $O/bench_c2_prof.json | cut -c1-300; tail -3 $O/pmc_traffic.log | cut -c1-600
