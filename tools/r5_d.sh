#!/bin/bash
# round-5 batch D: new tests + the 1-GPU cost of reserved CUs + the exposed-reduction report
O=gpurun_out/r5i; mkdir -p $O; rm -f $O/*
timeout 900 python -m pytest tests/test_gpu_conv.py tests/test_driver_extras.py -m gpu -x -q > $O/tests.log 2>&1; echo "pytest rc=$?" >> $O/tests.log
tail -4 $O/tests.log
for n in 0 8 16; do
  DASAC_SK_RESERVE_CUS=$n timeout 600 python bench.py --no-cpu-baseline --no-kernel-table --steps 5 --warmup 2 > $O/bench_res$n.json 2> $O/bench_res$n.err
done
DASAC_BENCH_DDP=1 timeout 600 python bench.py --no-cpu-baseline --no-kernel-table --steps 5 --warmup 2 > $O/bench_ddp.json 2> $O/bench_ddp.err
python - <<'PY'
import json
for f in ("bench_res0","bench_res8","bench_res16","bench_ddp"):
    try:
        l=json.load(open("gpurun_out/r5i/%s.json"%f)); d=l["config"]["distributed"]
        print(f, l["ms_per_step"], "reserved", d.get("reserved_cus"), d.get("per_rank"), d.get("wrapper"))
    except Exception as e: print(f, "failed", e)
PY
