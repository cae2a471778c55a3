#!/bin/bash
# Round 3, experiment 1: k-split wave grids of the direct-to-LDS GEMM tiles (OVQA_GEMM_KSPLIT bit 0 = 64x128 tier,
# bit 1 = 128x128 tier).  Correctness of the kernel suite under every setting, per-shape timing, step A/B.
set -o pipefail
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
python -m openvivqa_amd.build > /dev/null 2>&1 || { echo "library build failed"; exit 1; }
export OVQA_NO_BUILD=1
for ks in 3; do
  OVQA_GEMM_KSPLIT=$ks timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py tests/test_blocks_gpu.py -q -x -p no:cacheprovider -k "linear or block or ffn or layer" > gpurun_out/ksplit_tests_$ks.log 2>&1
  echo "tests ksplit=$ks exit $?"; tail -2 gpurun_out/ksplit_tests_$ks.log
done
for ks in 0 3; do
  echo "== gemm_bench ksplit=$ks"
  OVQA_GEMM_KSPLIT=$ks timeout -k 10 300 python scripts/gemm_bench.py fwd bwd_data 2>&1 | grep -v "^$" | cut -c1-400
done
for rep in 1 2; do
  for ks in 0 1 2 3; do
    OVQA_GEMM_KSPLIT=$ks timeout -k 10 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-roofline --repeats 3 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); print('ksplit=$ks', d['ms_per_step'], d['ms_per_step_min'], d['ms_per_step_max'])"
  done
done
