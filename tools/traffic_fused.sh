#!/bin/bash
# HBM traffic per launch of the FUSED student schedule (16-crop launches: the configuration rounds 2-5 quoted), two separate --pmc passes.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r6; mkdir -p $O; cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --fused --no-cpu-baseline --no-kernel-table --steps 1 --warmup 1"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/pf3 -o p -- $B > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/pf4 -o p -- $B > /dev/null 2>&1
python $R/tools/hbm_traffic.py /tmp/pf3/p_results.db /tmp/pf4/p_results.db > $O/hbm_traffic_fused.md
tail -8 $O/hbm_traffic_fused.md | cut -c1-160
