#!/bin/bash
# Regenerates the per-round measurement artefacts on an MI355X box (run through gpurun from the repo root):
#   gpurun_out/<tag>/bench.json            python bench.py (the driver's command)
#   gpurun_out/<tag>/kernel_stats.md       rocprofv3 --kernel-trace --stats of the same command (no CPU leg)
#   gpurun_out/<tag>/pmc_clock_mfma.txt    clock, matrix-pipe busy, VALU:MFMA per kernel (one --pmc pass)
#   gpurun_out/<tag>/hbm_traffic.md        FETCH_SIZE / WRITE_SIZE per launch (two separate --pmc passes)
# Copy what should be judged into profiles/ afterwards.  PMC passes carry --kernel-trace only (no sys/hip traces).
set -u
TAG=${1:-r6}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --no-cpu-baseline --no-kernel-table"
python $R/bench.py > $O/bench.json 2> $O/bench.err   # stdout = the one JSON line
rocprofv3 --kernel-trace --stats -d /tmp/p1 -o kt -- $B --steps 3 --warmup 1 > /dev/null 2>&1
python $R/tools/rocpd_stats.py /tmp/p1/kt_results.db > $O/kernel_stats.md
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA -d /tmp/p2 -o p -- $B --steps 1 --warmup 1 > /dev/null 2>&1
python $R/tools/pmc_clock.py /tmp/p2/p_results.db > $O/pmc_clock_mfma.txt
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/p3 -o p -- $B --steps 1 --warmup 1 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/p4 -o p -- $B --steps 1 --warmup 1 > /dev/null 2>&1
python $R/tools/hbm_traffic.py /tmp/p3/p_results.db /tmp/p4/p_results.db --json $O/traffic.json > $O/hbm_traffic.md
cat $O/bench.json
head -6 $O/kernel_stats.md | cut -c1-150
head -6 $O/pmc_clock_mfma.txt
cat $O/hbm_traffic.md | cut -c1-170
python $R/tools/step_shapes.py 2 > $O/step_shapes.txt 2>/dev/null; head -40 $O/step_shapes.txt
