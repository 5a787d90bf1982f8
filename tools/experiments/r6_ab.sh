#!/bin/bash
# Round-6 A/B on one box: the previous library (da-sac_amd/dasac_hip/libdasac_base.so, built from the round-5 kernels) against the
# current one -- weight gradients of the 3x3 layers (QTAP loader) and the whole step with / without the split-K tail.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r6_ab
mkdir -p "$O"
BASE=$R/da-sac_amd/dasac_hip/libdasac_base.so
cd $R
for rep in 1 2; do
  for shape in l3_3x3 l4_3x3 l3_1x1b; do
    DASAC_LIB=$BASE python tools/one_conv.py $shape wgrad 20 16 2>/dev/null | sed "s/^/base /"
    python tools/one_conv.py $shape wgrad 20 16 2>/dev/null | sed "s/^/new  /"
  done
done > $O/wgrad_ab.txt 2>&1
cat $O/wgrad_ab.txt
B="python bench.py --no-cpu-baseline --no-other-configs --steps 8 --warmup 2"
for rep in 1 2; do
  DASAC_LIB=$BASE $B > $O/bench_base_$rep.json 2> $O/bench_base_$rep.err
  DASAC_TAIL=0 $B > $O/bench_notail_$rep.json 2> $O/bench_notail_$rep.err
  $B > $O/bench_new_$rep.json 2> $O/bench_new_$rep.err
done
python - <<PY
import json,glob
for f in sorted(glob.glob("$O/bench_*.json")):
    try:
        d=json.load(open(f))
    except Exception as e:
        print(f, "unreadable", e); continue
    k=d["kernels"]
    print(f.split("/")[-1], d["ms_per_step"], d["ms_per_step_other_schedule"], {n:(k[n]["ms_per_step"],k[n].get("tflops")) for n in k if n.startswith("conv_")})
PY
