#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
timeout -k 10 420 python -m pytest tests/test_kernels_gpu.py -q -m gpu -p no:cacheprovider > gpurun_out/kernels.log 2>&1
echo "kernel tests exit $?"; tail -3 gpurun_out/kernels.log
timeout -k 10 420 python -m pytest tests/test_blocks_gpu.py -q -m gpu -p no:cacheprovider > gpurun_out/blocks.log 2>&1
echo "block tests exit $?"; grep -E "^E +Assertion" gpurun_out/blocks.log | cut -c1-600; tail -3 gpurun_out/blocks.log
timeout -k 10 420 python -m pytest tests/test_modules_gpu.py -q -m gpu -p no:cacheprovider > gpurun_out/modules.log 2>&1
echo "module tests exit $?"; grep -E "^(FAILED|ERROR)" gpurun_out/modules.log; tail -2 gpurun_out/modules.log
