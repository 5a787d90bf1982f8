#!/bin/bash
# HBM traffic (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes) of single layer shapes at the batch the fused student
# pass runs (16 crops): measured MB per launch next to the algorithmic MB.  Usage (GPU box): bash tools/traffic_shapes.sh > out.txt
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for cfg in "l3_1x1b fwd_res_bits" "l3_1x1b dgrad_res_bits" "l3_1x1a fwd" "l3_3x3 fwd" "l3_3x3 wgrad" "l3_1x1b wgrad" "l3_1x1a wgrad"; do
  set -- $cfg
  rm -rf /tmp/tf /tmp/tw
  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/tf -o p -- python $R/tools/one_conv.py $1 $2 6 16 > /tmp/tf.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/tw -o p -- python $R/tools/one_conv.py $1 $2 6 16 > /dev/null 2>&1
  grep "^done" /tmp/tf.log
  python $R/tools/hbm_traffic.py /tmp/tf/p_results.db /tmp/tw/p_results.db | grep -E "conv_gemm|conv_wgrad<" | cut -c1-200
done
