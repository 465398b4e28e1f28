#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
rocprofv3 --list-avail 2>/dev/null | grep -o "SQ_[A-Z_0-9]*" | sort -u | grep -E "WAIT|ACTIVE_INST|IFETCH|BARRIER|LEVEL_WAVES|THREAD_CYCLES" | tr '\n' ' ' > $O/pmc_avail2.txt; cat $O/pmc_avail2.txt; echo
run() { tag=$1; shift
  timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/pmcz_${tag} -- python $R/bench.py --steps 3 --warmup 1 --no-graph --no-cpu-baseline > /dev/null 2> $O/pmcz_${tag}.err || tail -3 $O/pmcz_${tag}.err; }
run a SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS
run b SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_VMEM
cd $R
python tools/pmc_counters.py $O/pmc_waits.txt "default:$O/pmcz_a,$O/pmcz_b" -- "stem_stream_kernel" "conv_tap_kernel<step::bf16_t, 3, 3" "conv_tap_kernel<step::bf16_t, 0, 1, 3, 3, 3, 2, 2, 8" > /dev/null
cat $O/pmc_waits.txt | cut -c1-130
rm -rf $O/pmcz_*
