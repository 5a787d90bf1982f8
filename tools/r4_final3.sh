#!/bin/bash
# the round's last tree: GPU suite twice more + smoke + the driver's bench command
O=gpurun_out/r4_final3; mkdir -p $O
for i in 1 2; do
  python -m pytest tests -m gpu -x -q > $O/run$i.log 2>&1; echo "run $i rc=$? :: $(tail -1 $O/run$i.log)" >> $O/summary.txt
done
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$? :: $(tail -2 $O/smoke.log | head -1)" >> $O/summary.txt
python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" >> $O/summary.txt
cat $O/summary.txt
python - <<'PY'
import json
l=json.load(open("gpurun_out/r4_final3/bench.json"))
print(l["value"], l["ms_per_step"], l["value_train_py_api"], l["roofline"]["frac"], {k:v.get("ms_per_step") for k,v in l["other_configs"].items()})
print({k:(v["ms_per_step"], v.get("algorithmic_TB_per_s")) for k,v in l["kernels"].items() if not k.startswith("conv")})
PY
