#!/bin/bash
O=gpurun_out/r4_final2; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; echo "pytest rc=$?" >> $O/tests.log
tail -2 $O/tests.log
for k in WORLD_SIZE RANK LOCAL_RANK MASTER_ADDR MASTER_PORT; do unset $k; done
DASAC_BENCH_RANKS_PER_GPU=2 python bench.py --gpus 2 --steps 4 --warmup 1 --profile-steps 1 > $O/two_ranks_fullsize.json 2> $O/two_ranks_fullsize.err; echo "2-rank bench rc=$?"
python - <<'PY'
import json
l=json.load(open("gpurun_out/r4_final2/two_ranks_fullsize.json"))
print(l["n_gpus"], l["value"], l["ms_per_step"], l["config"]["distributed"])
PY
