#!/bin/bash
# PMC passes over tools/head_bw.py (SAC-head / pointwise kernels at the cfg-3 shape): HBM bytes, L2 hit rate, wave stall share.
# Usage (GPU box, via gpurun): bash tools/head_pmc.sh <outdir>.   Counter passes carry --kernel-trace only.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/${1:-gpurun_out/head_pmc}
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
i=0
MAXP=${HEAD_PMC_PASSES:-5}     # 2 = HBM bytes only (FETCH_SIZE, WRITE_SIZE)
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TA_BUSY_avr GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  [ $i -gt $MAXP ] && break
  rocprofv3 --kernel-trace --pmc $set -d /tmp/hp$i -o p -- python $R/tools/head_bw.py > $O/pass$i.log 2>&1
  python $R/tools/pmc_table.py /tmp/hp$i/p_results.db > $O/pass$i.md 2>> $O/pass$i.log
done
cat $O/pass*.md
