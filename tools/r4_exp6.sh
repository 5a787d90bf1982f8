#!/bin/bash
# NOTE: needs a second library built with -DDASAC_SK_PER_CU=4 at da-sac_amd/dasac_hip/libdasac_hip_sk4.so (see profiles/r4_streamk_4_workers_experiment.txt).
# stream-K at FOUR workers per CU (128 registers, spills) against the shipped three (168 registers): whole-step effect
O=gpurun_out/r4h; mkdir -p $O
R=$PWD
for v in base sk4; do
  if [ $v = sk4 ]; then export DASAC_LIB=$R/da-sac_amd/dasac_hip/libdasac_hip_sk4.so; else unset DASAC_LIB; fi
  python bench.py --no-cpu-baseline --no-other-configs --steps 6 --warmup 2 > $O/cfg3_$v.json 2> $O/cfg3_$v.err
  python bench.py --config cfg2 --no-cpu-baseline --steps 10 --warmup 2 > $O/cfg2_$v.json 2> /dev/null
done
python - <<'PY'
import json
for v in ("base","sk4"):
    for c in ("cfg3","cfg2"):
        try:
            l=json.load(open("gpurun_out/r4h/%s_%s.json"%(c,v)))
            k=l["kernels"]
            print(v, c, l["ms_per_step"], {n:(k[n]["ms_per_step"],k[n].get("tflops")) for n in k if n.startswith("conv")})
        except Exception as e: print(v,c,"failed",e)
PY
