# HBM fetch per launch of one layer shape under two settings (rocprofv3 --pmc FETCH_SIZE; raw KiB, x2 on gfx950)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for m in fwd wgrad; do for p in fp32 bf16x3; do
DASAC_PRECISION=$p rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/f$m$p -o p -- python $R/tools/one_conv.py l3_3x3 $m 6 > /dev/null 2>&1
echo "$m $p"; python $R/tools/pmc_summary.py /tmp/f$m$p/p_results.db conv_ | cut -c1-60,75-
done; done
