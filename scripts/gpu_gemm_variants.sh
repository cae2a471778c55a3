#!/bin/bash
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
for v in ${VARIANTS:-12 13}; do
  echo "== variant $v: correctness"
  OVQA_GEMM_VARIANT=$v timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py -q -m gpu -p no:cacheprovider -k "linear or grouped" 2>&1 | tail -1
  echo "== variant $v: bench"
  OVQA_GEMM_VARIANT=$v timeout -k 10 300 python scripts/gemm_bench.py fwd bwd_data 2>&1 | grep -v amdgpu.ids | tee gpurun_out/gemm_variant_$v.log
done
