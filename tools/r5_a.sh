#!/bin/bash
# round-5 measurement batch A: epilogue without store round trips (conv_gemm)
O=gpurun_out/r5a; mkdir -p $O
python -m pytest tests/test_gpu_conv.py -m gpu -x -q --durations=5 > $O/tests.log 2>&1; echo "pytest rc=$?" >> $O/tests.log
for cfg in "l3_1x1b fwd" "l3_1x1b fwd_res" "l3_1x1b fwd_res_bits" "l3_1x1a dgrad_res_bits" "l3_1x1a dgrad_res_mask" "l3_1x1a fwd" "l3_3x3 fwd" "l4_3x3 fwd"; do
  set -- $cfg
  python tools/one_conv.py $1 $2 20 16 2>&1 | grep "^done" >> $O/one_conv.txt
done
python bench.py --no-cpu-baseline --steps 5 --warmup 2 > $O/bench.json 2> $O/bench.err
tail -5 $O/tests.log; cat $O/one_conv.txt
python - <<'PY'
import json
l=json.load(open("gpurun_out/r5a/bench.json")); print("ms_per_step", l["ms_per_step"], "value", l["value"], "roofline", l.get("roofline")); print({k:(v.get("ms_per_step") if isinstance(v,dict) else v) for k,v in l.get("other_configs",{}).items()})
PY
