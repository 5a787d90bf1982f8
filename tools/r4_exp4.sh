#!/bin/bash
O=gpurun_out/r4f; mkdir -p $O
python -m pytest tests/test_gpu_conv.py tests/test_gpu_bn_train.py tests/test_gpu_zz_determinism.py -m gpu -x -q -k "statistics or bn or baseline or golden" > $O/tests.log 2>&1; echo "pytest rc=$?" >> $O/tests.log
python -m pytest tests/test_gpu_fullres.py -m gpu -x -q -k "cfg2" > $O/tests2.log 2>&1; echo "pytest rc=$?" >> $O/tests2.log
tail -3 $O/tests.log; tail -3 $O/tests2.log
bash tools/r4_profile.sh
