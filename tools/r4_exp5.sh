#!/bin/bash
O=gpurun_out/r4g; mkdir -p $O
python -m pytest tests/test_gpu_head.py tests/test_gpu_models.py tests/test_gpu_fullres.py -m gpu -x -q -k "refine or golden or warp or head_parity or full_size or sac_steps or pool" > $O/tests.log 2>&1; echo "pytest rc=$?" >> $O/tests.log
python tools/head_bw.py > $O/head_bw.txt 2>&1
tail -4 $O/tests.log; cat $O/head_bw.txt
