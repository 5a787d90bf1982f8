#!/bin/bash
# NOTE: the DASAC_EXP_BM64_KSTEPS switch this script drives existed only for the measurement (profiles/r4_bm64_occupancy5_experiment.txt);
# it was removed again, so the bm64_* runs now equal "base".
# round-4 measurement batch 1 (run through gpurun from the repo root)
O=gpurun_out/r4c; mkdir -p $O
python -m pytest tests/test_gpu_conv.py tests/test_gpu_bn_train.py tests/test_gpu_head.py tests/test_gpu_zz_determinism.py -m gpu -x -q --durations=8 > $O/tests.log 2>&1; echo "pytest rc=$?" >> $O/tests.log
python -m pytest tests/test_gpu_fullres.py -m gpu -x -q -k "cfg2 or head_parity" > $O/tests2.log 2>&1; echo "pytest rc=$?" >> $O/tests2.log
python bench.py --config cfg2 --no-cpu-baseline --steps 10 --warmup 2 > $O/cfg2.json 2> $O/cfg2.err
DASAC_GEMM_STATS=0 python bench.py --config cfg2 --no-cpu-baseline --steps 10 --warmup 2 > $O/cfg2_nostats.json 2> /dev/null
DASAC_EXP_B=16 python tools/gemm_exp.py base > $O/gemm_exp.txt 2>&1
DASAC_EXP_B=16 DASAC_EXP_BM64_KSTEPS=16 python tools/gemm_exp.py bm64_k16 >> $O/gemm_exp.txt 2>&1
DASAC_EXP_B=16 DASAC_EXP_BM64_KSTEPS=64 python tools/gemm_exp.py bm64_k64 >> $O/gemm_exp.txt 2>&1
python tools/head_bw.py > $O/head_bw.txt 2>&1
tail -4 $O/tests.log; tail -3 $O/tests2.log; cat $O/gemm_exp.txt; cat $O/head_bw.txt
python - <<'PY'
import json
for f in ("cfg2","cfg2_nostats"):
    try:
        l=json.load(open("gpurun_out/r4c/%s.json"%f)); print(f, l["ms_per_step"], l["value"])
    except Exception as e: print(f, "failed", e)
PY
