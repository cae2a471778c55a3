#!/bin/bash
export PYTHONDONTWRITEBYTECODE=1
timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py -q -m gpu -p no:cacheprovider -k "linear or grouped" 2>&1 | tail -2
for t in 0 96; do echo "== small-tile threshold $t"; OVQA_GEMM_SMALL_TILES=$t timeout -k 10 300 python scripts/gemm_bench.py fwd bwd_data 2>&1 | grep -v amdgpu.ids | grep '"M": 1280'; done
