#!/bin/bash
# Round-4 evidence batch (gpurun, repo root): headline bench + rocprofv3 kernel trace + PMC passes + per-shape table,
# HBM bytes of the head kernels, cfg-2 kernel trace after the BN work.
R=${GRAFT_REPO_ROOT:-$(pwd)}
bash tools/profile_round.sh r4 > gpurun_out/r4_profile_round.log 2>&1
HEAD_PMC_PASSES=2 bash tools/head_pmc.sh gpurun_out/r4/head_pmc > gpurun_out/r4_head_pmc.log 2>&1
python tools/head_bw.py > gpurun_out/r4/head_bw.txt 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/pc2 -o kt -- python $R/bench.py --config cfg2 --no-cpu-baseline --no-kernel-table --steps 4 --warmup 1 > $R/gpurun_out/r4/cfg2_bench.json 2>/dev/null
python $R/tools/rocpd_stats.py /tmp/pc2/kt_results.db > $R/gpurun_out/r4/cfg2_kernel_stats.md
cd $R
tail -5 gpurun_out/r4_profile_round.log | cut -c1-200; head -12 gpurun_out/r4/cfg2_kernel_stats.md | cut -c1-130; cat gpurun_out/r4/head_bw.txt
