cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export DASAC_PRECISION=bf16x3
for m in fwd wgrad; do
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY -d /tmp/q1$m -o p -- python $R/tools/one_conv.py l3_3x3 $m 6 > /dev/null 2>&1
python $R/tools/pmc_summary.py /tmp/q1$m/p_results.db conv_ | cut -c1-40,51-
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VMEM -d /tmp/q2$m -o p -- python $R/tools/one_conv.py l3_3x3 $m 6 > /dev/null 2>&1
python $R/tools/pmc_summary.py /tmp/q2$m/p_results.db conv_ | cut -c1-40,51-
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/q3$m -o p -- python $R/tools/one_conv.py l3_3x3 $m 6 > /dev/null 2>&1
python $R/tools/pmc_summary.py /tmp/q3$m/p_results.db conv_ | cut -c1-40,51-
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA -d /tmp/q4$m -o p -- python $R/tools/one_conv.py l3_3x3 $m 6 > /dev/null 2>&1
python $R/tools/pmc_summary.py /tmp/q4$m/p_results.db conv_ | cut -c1-40,51-
done
