#!/bin/bash
# NOTE: the variant libraries (-DDASAC_WG_MINPIX=512 / 256) existed for the measurement only; 256 is the shipped default now.
# weight-gradient pixel splits at small batch: at least 1024 (shipped) / 512 / 256 pixels per split (variant libraries via DASAC_LIB)
O=gpurun_out/r4k; mkdir -p $O
R=$PWD
for v in base wg512 wg256; do
  if [ $v = base ]; then unset DASAC_LIB; else export DASAC_LIB=$R/da-sac_amd/dasac_hip/libdasac_hip_$v.so; fi
  python bench.py --config cfg2 --no-cpu-baseline --steps 10 --warmup 2 > $O/cfg2_$v.json 2>/dev/null
done
export DASAC_LIB=$R/da-sac_amd/dasac_hip/libdasac_hip_wg256.so
python -m pytest tests/test_gpu_conv.py tests/test_gpu_bn_train.py -m gpu -x -q > $O/tests_wg256.log 2>&1; echo "pytest(wg256) rc=$?"; tail -1 $O/tests_wg256.log
python - <<'PY'
import json
for v in ("base","wg512","wg256"):
    try:
        l=json.load(open("gpurun_out/r4k/cfg2_%s.json"%v)); k=l["kernels"]
        print(v, l["ms_per_step"], {n:(k[n]["ms_per_step"],k[n].get("tflops"),k[n]["launches_per_step"]) for n in k if n.startswith("conv")})
    except Exception as e: print(v,"failed",e)
PY
