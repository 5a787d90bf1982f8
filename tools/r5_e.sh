#!/bin/bash
# round-5 batch E: 4 GiB window -- the >2 GiB test, the conv suite, cfg-5 fused vs two passes
O=gpurun_out/r5j; mkdir -p $O; rm -f $O/*
timeout 900 python -m pytest tests/test_gpu_conv.py -m gpu -x -q --durations=4 > $O/tests.log 2>&1; echo "pytest rc=$?" >> $O/tests.log
tail -8 $O/tests.log
timeout 600 python bench.py --config cfg5 --no-cpu-baseline --steps 5 --warmup 2 > $O/cfg5.json 2> $O/cfg5.err
timeout 600 python bench.py --config cfg5 --two-pass --no-cpu-baseline --no-kernel-table --steps 5 --warmup 2 > $O/cfg5_two.json 2> $O/cfg5_two.err
python - <<'PY'
import json
for f in ("cfg5","cfg5_two"):
    try:
        l=json.load(open("gpurun_out/r5j/%s.json"%f)); print(f, l["ms_per_step"], l["value"], l["config"]["student_schedule"][:40], l.get("ms_per_step_other_schedule"), l["check"])
    except Exception as e: print(f, "failed", e, open("gpurun_out/r5j/%s.err"%f).read()[-600:])
PY
