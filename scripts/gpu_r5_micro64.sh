#!/bin/bash
# round 5: tile tiers of the question stack's GEMMs (1280 activation rows): tests with the new forms on, then per-kernel A/B
export OVQA_NO_BUILD=1 PYTHONDONTWRITEBYTECODE=1
mkdir -p gpurun_out
for cfg in "OVQA_GEMM_MICRO64=4 OVQA_GEMM_TINY_MAXR=1024" "OVQA_GEMM_MICRO64=6"; do
  env $cfg timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py tests/test_blocks_gpu.py -x -q -p no:cacheprovider -k "linear or gemm or ffn or block" 2>&1 | tail -2 || exit 1
done
bash scripts/gpu_kernel_ab.sh "gemm_bf16_glds" "-" "OVQA_GEMM_TINY_MAXR=1024" "OVQA_GEMM_TINY_MAXR=1024 OVQA_GEMM_MICRO64=4" "OVQA_GEMM_TINY_MAXR=1024 OVQA_GEMM_MICRO64=6"
