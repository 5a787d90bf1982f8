#!/bin/bash
# round-5 batch C: K-loop instruction diet of conv_gemm -- correctness + single-shape timings + bench
O=gpurun_out/r5f; mkdir -p $O; rm -f $O/*
timeout 900 python -m pytest tests/test_gpu_conv.py -m gpu -x -q > $O/tests.log 2>&1; echo "pytest rc=$?" >> $O/tests.log
tail -5 $O/tests.log
for cfg in "l3_1x1b fwd" "l3_1x1b fwd_res" "l3_1x1b fwd_res_bits" "l3_1x1a dgrad_res_bits" "l3_1x1a fwd" "l3_3x3 fwd" "l4_3x3 fwd"; do
  set -- $cfg
  DASAC_MSWEEP=0 timeout 300 python tools/one_conv.py $1 $2 20 16 2>&1 | grep "^done" >> $O/one_conv.txt
done
cat $O/one_conv.txt
DASAC_MSWEEP=0 timeout 600 python bench.py --no-cpu-baseline --steps 5 --warmup 2 > $O/bench.json 2> $O/bench.err
python - <<'PY'
import json
l=json.load(open("gpurun_out/r5f/bench.json")); print("ms_per_step", l["ms_per_step"], "value", l["value"], "roofline frac", l["roofline"]["frac"], l["roofline"]["achieved"])
for k,v in l.get("kernels",{}).items():
    if "conv" in k or "gemm" in k: print(k, v)
PY
