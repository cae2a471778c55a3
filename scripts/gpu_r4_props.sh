#!/bin/bash
export PYTHONDONTWRITEBYTECODE=1 OVQA_NO_BUILD=1
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_modules_gpu.py -q -m gpu -p no:cacheprovider -x -k "baseline_size_backward_is_linear or baseline_size_padding" > gpurun_out/tests_props.log 2>&1 || { echo "tests failed"; grep -E "^(FAILED|ERROR)|^E  " gpurun_out/tests_props.log | head -20; exit 1; }
tail -1 gpurun_out/tests_props.log
