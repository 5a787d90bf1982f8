#!/bin/bash
# last tree of the round (weight-gradient splits down to 256 pixels): GPU suite, smoke, the driver's bench command
O=gpurun_out/r4_final4; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; echo "suite rc=$? :: $(tail -1 $O/tests.log)" > $O/summary.txt
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/summary.txt
python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" >> $O/summary.txt
cat $O/summary.txt
python - <<'PY'
import json
l=json.load(open("gpurun_out/r4_final4/bench.json"))
print(l["value"], l["ms_per_step"], l["value_train_py_api"], l["roofline"]["frac"], l["other_configs"])
print({k:(v["ms_per_step"], v.get("tflops")) for k,v in l["kernels"].items() if k.startswith("conv")})
PY
