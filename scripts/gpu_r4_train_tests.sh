#!/bin/bash
export PYTHONDONTWRITEBYTECODE=1 OVQA_NO_BUILD=1
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_train_gpu.py -q -m gpu -p no:cacheprovider -x > gpurun_out/tests_train.log 2>&1 || { echo "tests failed"; grep -E "^(FAILED|ERROR)" gpurun_out/tests_train.log | head; tail -40 gpurun_out/tests_train.log; exit 1; }
tail -1 gpurun_out/tests_train.log


