cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for sk in 0 1; do
DASAC_STREAMK=$sk rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/f$sk -o p -- python $R/tools/one_conv.py l3_3x3 fwd 6 > /dev/null 2>&1
echo "STREAMK=$sk"; python $R/tools/pmc_summary.py /tmp/f$sk/p_results.db conv_gemm | cut -c1-60,75-
done
