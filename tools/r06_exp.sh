R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd /tmp; export TMPDIR=/tmp
i=0
for grp in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD" "SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $O/pmc_pw_$i -- python $R/tools/ab_bench.py --batch 136 --rounds 1 --iters 2 --custom a832_1024,832,1024,1,9,7,7 --custom c1024_256,1024,256,1,9,7,7 --var conv_pws=0 --var conv_pws=0,conv_nb=3 > /dev/null 2> $O/pmc_pw_$i.err
done
cd $R
python tools/pmc_dump.py conv_pw_kernel $O/pmc_pw_1 $O/pmc_pw_2 $O/pmc_pw_3 $O/pmc_pw_4 > $O/pmc_pw.txt 2>&1
rm -rf $O/pmc_pw_1 $O/pmc_pw_2 $O/pmc_pw_3 $O/pmc_pw_4
cat $O/pmc_pw.txt
