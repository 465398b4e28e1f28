#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
STEP_CONV_PHASED=0 timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d $O/pmcx_a0 -- python $R/bench.py --steps 3 --warmup 1 --no-graph --no-cpu-baseline > /dev/null 2> $O/pmcx_a0.err || tail -3 $O/pmcx_a0.err
cd $R
python tools/pmc_counters.py $O/pmc_3x3x3_classic_a.txt "classic=$O/pmcx_a0" -- "conv_tap_kernel<step::bf16_t, 3, 3, 3, 3, 3" "conv_tap_kernel<step::bf16_t, 0, 3, 3, 3, 3" "conv_tap_kernel<step::bf16_t, 0, 2, 3, 3, 3" "conv_tap_kernel<step::bf16_t, 0, 1, 3, 3, 3, 2, 2, 8" | cut -c1-170
rm -rf $O/pmcx_*
