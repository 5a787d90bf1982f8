#!/bin/bash
# round-5 batch B: M-sweep kernel correctness + single-shape timings (+ the same shapes with DASAC_MSWEEP=0)
O=gpurun_out/r5e; mkdir -p $O; rm -f $O/*
timeout 600 python -m pytest tests/test_gpu_conv.py -m gpu -x -q -k "msweep" > $O/tests.log 2>&1; echo "pytest rc=$?" >> $O/tests.log
tail -15 $O/tests.log
for cfg in "l3_1x1b fwd" "l3_1x1b fwd_res" "l3_1x1b fwd_res_bits" "l3_1x1a dgrad_res_bits"; do
  set -- $cfg
  timeout 300 python tools/one_conv.py $1 $2 20 16 2>&1 | grep "^done" >> $O/one_conv.txt
  DASAC_MSWEEP=0 timeout 300 python tools/one_conv.py $1 $2 20 16 2>&1 | grep "^done" | sed 's/^done/base/' >> $O/one_conv.txt
done
cat $O/one_conv.txt
