#!/bin/bash
# round-3 switches: the GPU kernel / block / module suites under every A/B path added this round
mkdir -p gpurun_out
python -m openvivqa_amd.build > /dev/null 2>&1 || exit 1
export OVQA_NO_BUILD=1
for e in "OVQA_FORCE_SIMPLE=1" "OVQA_NO_FUSED_QKV=1" "OVQA_NO_FUSED_Q=1" "OVQA_NO_FUSED_DO=1" "OVQA_QATT_PAIR=0" "OVQA_QATT_NBUF=2" "OVQA_GEMM_BIG16=0" "OVQA_GEMM_SKINNY_MAXROWS=0" "OVQA_GEMM_KSPLIT=3 OVQA_GEMM_KSPLIT_MINK=512 OVQA_DW_KSPLIT=1" "OVQA_DECODE_SPLIT_MIN=1000"; do
  echo "== $e"
  env $e timeout -k 10 500 python -m pytest tests/test_kernels_gpu.py tests/test_blocks_gpu.py tests/test_modules_gpu.py -q -x -m gpu 2>&1 | tail -1
done
