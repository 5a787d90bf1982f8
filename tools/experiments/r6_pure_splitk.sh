#!/bin/bash
# Round-6 experiment: small-batch launches (fewer tiles than resident blocks) as pure split-K pieces of the tile kernel (4 blocks per CU)
# instead of the persistent stream-K kernel (3 per CU).  DASAC_PURE_SPLITK=1 switches it on in this build.
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r6_pure; mkdir -p $O; cd $R
DASAC_PURE_SPLITK=1 python -m pytest tests/test_gpu_conv.py tests/test_gpu_bn_train.py -m gpu -x -q 2>&1 | tail -3
for rep in 1 2; do
  for v in 0 1; do
    DASAC_PURE_SPLITK=$v python bench.py --config cfg2 --no-cpu-baseline --no-other-configs --steps 10 --warmup 2 > $O/cfg2_$v_$rep.json 2>/dev/null
    python - <<PY
import json; d=json.load(open("$O/cfg2_$v_$rep.json")); k=d["kernels"]
print("cfg2 pure=$v rep=$rep", d["ms_per_step"], {n:(k[n]["ms_per_step"],k[n].get("tflops")) for n in k if n.startswith("conv_")})
PY
  done
done
for v in 0 1; do
  DASAC_PURE_SPLITK=$v python bench.py --no-cpu-baseline --no-other-configs --steps 8 --warmup 2 > $O/cfg3_$v.json 2>/dev/null
  python - <<PY
import json; d=json.load(open("$O/cfg3_$v.json")); k=d["kernels"]
print("cfg3 pure=$v", d["ms_per_step"], d["ms_per_step_other_schedule"], {n:(k[n]["ms_per_step"],k[n].get("tflops")) for n in k if n.startswith("conv_")})
PY
done
