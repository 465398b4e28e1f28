#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
run() { tag=$1; shift
  timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/pmcy_${tag} -- python $R/bench.py --steps 3 --warmup 1 --no-graph --no-cpu-baseline > /dev/null 2> $O/pmcy_${tag}.err || tail -3 $O/pmcy_${tag}.err; }
run a SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES
run b SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA
run c SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS
cd $R
python tools/pmc_counters.py $O/pmc_others.txt "default:$O/pmcy_a,$O/pmcy_b,$O/pmcy_c" -- "stem_stream_kernel" "conv_pw_kernel" "conv_pws_kernel" "maxpool_sep_kernel" "conv_igemm_kernel" "2, 2, 4, 0>" > /dev/null
sed -i 's/=/=/' $O/pmc_others.txt
grep -E "^\[|BANK_CONFLICT|IDX_ACTIVE|BUSY_CU|MFMA_BUSY|MFMA busy" $O/pmc_others.txt | cut -c1-150
rm -rf $O/pmcy_*
