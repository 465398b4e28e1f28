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
prof c2 --steps 100 --warmup 10
prof c3 --config c3 --steps 20 --warmup 5 --no-cpu-baseline
prof c4_bf16 --config c4 --dtype bf16 --steps 10 --warmup 3 --no-cpu-baseline
prof c4_f32 --config c4 --steps 10 --warmup 3 --no-cpu-baseline
prof c5 --config c5 --steps 20 --warmup 5 --no-cpu-baseline
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$c -- python $R/bench.py --steps 4 --warmup 2 --no-graph --no-cpu-baseline > $O/pmc_$c.json 2> $O/pmc_$c.err
done
cd $R
K1='void step::conv_tap_kernel<step::bf16_t, 3, 3, 3, 3, 3, 2, 2, 8, 1>(step::ConvParams)'
K2='void step::stem_stream_kernel<step::bf16_t, 2>(step::StemParams)'
K3='void step::conv_tap_kernel<step::bf16_t, 0, 2, 3, 3, 3, 2, 2, 8, 1>(step::ConvParams)'
K4='void step::conv_tap_kernel<step::bf16_t, 0, 3, 3, 3, 3, 2, 2, 8, 1>(step::ConvParams)'
K5='void step::conv_tap_kernel<step::bf16_t, 0, 1, 3, 3, 3, 2, 2, 8, 1>(step::ConvParams)'
K6='void step::maxpool_sep_kernel<step::bf16_t, 3, 3, 3, 1, 1, 1, 256>(step::bf16_t const*, step::bf16_t*, step::PoolParams, int, int, int, int, int, int, int, int)'
K7='void step::maxpool_sep_kernel<step::bf16_t, 1, 3, 3, 1, 2, 2, 256>(step::bf16_t const*, step::bf16_t*, step::PoolParams, int, int, int, int, int, int, int, int)'
K8='void step::conv_pw_kernel<step::bf16_t, 1, 8>(step::ConvParams)'
K9='void step::conv_pws_kernel<step::bf16_t, 3, 4>(step::ConvParams, int)'
K10='void step::conv_pw_kernel<step::bf16_t, 3, 4>(step::ConvParams)'
K11='void step::conv_tap_kernel<step::bf16_t, 3, 1, 3, 3, 3, 2, 2, 8, 1>(step::ConvParams)'    # conv3d_2c's partial last round (C2)
python tools/pmc_traffic.py $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/traffic_latest.json "$K1" "$K2" "$K3" "$K4" "$K5" "$K6" "$K7" "$K8" "$K9" "$K10" "$K11" > $O/pmc_traffic.log 2>&1
python - <<P
import json
f='$O/traffic_latest.json'; j=json.load(open(f)); k=j['kernels']
if '''$K11''' in k: k['''$K11''']['with']='''$K1'''
json.dump(j, open(f,'w'), indent=1)
P
rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE
head -22 $O/prof_c2_summary.txt | cut -c1-200; cat $O/bench_c2_prof.json | cut -c1-300; tail -3 $O/pmc_traffic.log | cut -c1-600
