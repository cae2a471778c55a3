#!/bin/bash
# round 5: direct-to-LDS image staging in the role-split attention backward: tests in both forms, then per-kernel A/B
export OVQA_NO_BUILD=1 PYTHONDONTWRITEBYTECODE=1
mkdir -p gpurun_out
for cfg in "OVQA_ROLES_DMA=1" "OVQA_ROLES_DMA=1 OVQA_DOBWD_ROLES=0"; do
  env $cfg timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py tests/test_blocks_gpu.py -x -q -p no:cacheprovider -k "attention or attn or block or mha" 2>&1 | tail -2 || exit 1
done
bash scripts/gpu_kernel_ab.sh "attn_bwd_roles" "OVQA_ROLES_DMA=0" "OVQA_ROLES_DMA=1" "OVQA_ROLES_DMA=0 OVQA_DOBWD_ROLES=0" "OVQA_ROLES_DMA=1 OVQA_DOBWD_ROLES=0"
