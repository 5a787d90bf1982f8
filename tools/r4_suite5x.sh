#!/bin/bash
# the driver's GPU suite five times back to back on ONE box, zero failures expected.
O=gpurun_out/r4_5x; mkdir -p $O
echo "host $(hostname) $(date -u +%FT%TZ) commit-tree $(md5sum da-sac_amd/dasac_hip/libdasac_hip.so | cut -c1-12)" > $O/summary.txt
for i in 1 2 3 4 5; do
  s=$(date +%s)
  python -m pytest tests -m gpu -x -q > $O/run$i.log 2>&1
  rc=$?
  e=$(date +%s)
  echo "run $i: rc=$rc wall=$((e-s))s :: $(tail -1 $O/run$i.log)" >> $O/summary.txt
done
cat $O/summary.txt
