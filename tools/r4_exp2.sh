#!/bin/bash
O=gpurun_out/r4d; mkdir -p $O
python -m pytest tests/test_gpu_head.py tests/test_gpu_models.py -m gpu -x -q --durations=5 > $O/tests.log 2>&1; echo "pytest rc=$?" >> $O/tests.log
python tools/head_bw.py > $O/head_bw.txt 2>&1
tail -12 $O/tests.log; cat $O/head_bw.txt
