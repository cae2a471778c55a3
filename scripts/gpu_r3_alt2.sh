#!/bin/bash
mkdir -p gpurun_out
python -m openvivqa_amd.build > /dev/null 2>&1 || exit 1
export OVQA_NO_BUILD=1
OVQA_FORCE_SIMPLE=1 timeout -k 10 500 python -m pytest tests/test_kernels_gpu.py tests/test_blocks_gpu.py tests/test_modules_gpu.py -q -x -m gpu > gpurun_out/alt_simple.log 2>&1; grep -E "^E |^FAILED|Error" gpurun_out/alt_simple.log | head -12
OVQA_GEMM_KSPLIT=3 OVQA_GEMM_KSPLIT_MINK=512 OVQA_DW_KSPLIT=1 timeout -k 10 500 python -m pytest tests/test_kernels_gpu.py tests/test_blocks_gpu.py tests/test_modules_gpu.py -q -x -m gpu > gpurun_out/alt_ksplit.log 2>&1; grep -E "^E |^FAILED|Error" gpurun_out/alt_ksplit.log | head -12
