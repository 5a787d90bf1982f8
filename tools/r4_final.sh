#!/bin/bash
O=gpurun_out/r4_final; mkdir -p $O
python -m pytest tests -m gpu -x -q --durations=12 > $O/tests.log 2>&1; echo "pytest rc=$?" >> $O/tests.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log
python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
tail -4 $O/tests.log; tail -2 $O/smoke.log
python - <<'PY'
import json
l=json.load(open("gpurun_out/r4_final/bench.json"))
print(l["value"], l["ms_per_step"], l["value_train_py_api"], l["roofline"]["frac"], {k:v.get("ms_per_step") for k,v in l["other_configs"].items()})
PY
