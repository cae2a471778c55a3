#!/bin/bash
# The GPU suite on the alternate kernel paths (VALU reference kernels, register-staged GEMM, two-kernel attention
# backward): the paths unusual shapes fall back to must stay green.
for env in "OVQA_FORCE_SIMPLE=1" "OVQA_GEMM_VARIANT=0" "OVQA_ATTN_BWD_MERGED=0" "OVQA_DEFER_WGRAD=0" "OVQA_DW_GLDS=0" \
           "OVQA_NO_FUSED_QKV=1" "OVQA_GEMM_MICRO_TILES=0" "OVQA_QKV_FORM=0" "OVQA_QKV_FORM=2"; do
  echo "== $env"
  env $env python -m pytest tests/test_kernels_gpu.py tests/test_blocks_gpu.py tests/test_modules_gpu.py -q -x 2>&1 | tail -1
done
