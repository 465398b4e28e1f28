#!/bin/bash
# round 4: SQ counters of the C2 step's kernels and of the training step's weight-gradient / pool-backward kernels (separate --pmc passes,
# --kernel-trace only beside them)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
i=0
for grp in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD"; do
  i=$((i+1))
  timeout 400 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $O/sq2_$i -- python $R/bench.py --steps 4 --warmup 2 --no-graph --no-cpu-baseline --sustained-seconds 0 > /dev/null 2> $O/sq2_$i.err
  timeout 400 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $O/sq4_$i -- python $R/bench.py --config c4 --dtype bf16 --clips 8 --tubes 15 --steps 2 --warmup 1 --no-graph --no-cpu-baseline > /dev/null 2> $O/sq4_$i.err
done
cd $R
{ echo "# rocprofv3 --pmc (three separate passes) of bench.py --steps 4 --warmup 2 --no-graph (C2, bf16, 8 clips): per-launch means; MFMA busy share = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x SQ_BUSY_CU_CYCLES)"
  for k in stem_stream_kernel conv_tap_pre_kernel conv_tap_group_pw_kernel conv_tap_group_kernel pool333_pw_kernel; do python tools/pmc_dump.py $k $O/sq2_1 $O/sq2_2 $O/sq2_3; done; } > $O/r04_pmc_c2.txt
{ echo "# rocprofv3 --pmc (three separate passes) of bench.py --config c4 --dtype bf16 --clips 8 --tubes 15 --steps 2 --warmup 1 --no-graph: per-launch means"
  for k in conv_wgrad16_lds12 conv_wgrad16_pws_kernel maxpool333_bwd_kernel roi_align_bwd_gather_nhwc act_grad8; do python tools/pmc_dump.py $k $O/sq4_1 $O/sq4_2 $O/sq4_3; done; } > $O/r04_pmc_c4.txt
rm -rf $O/sq2_* $O/sq4_*
head -40 $O/r04_pmc_c2.txt
