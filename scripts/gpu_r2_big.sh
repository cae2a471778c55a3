#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py -q -m gpu -p no:cacheprovider -k "linear" > gpurun_out/tests_big.log 2>&1
echo "linear kernel tests exit $?"; grep -E "^(FAILED|ERROR)" gpurun_out/tests_big.log | head; tail -2 gpurun_out/tests_big.log
for big in 0 1; do
  OVQA_GEMM_BIG=$big timeout -k 10 200 python scripts/gemm_bench.py fwd bwd_data > gpurun_out/gemm_bench_big$big.log 2>&1; echo "gemm bench big=$big exit $?"
  cut -c1-300 gpurun_out/gemm_bench_big$big.log | grep -v amdgpu.ids
done
for big in 0 1; do
  OVQA_GEMM_BIG=$big timeout -k 10 200 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-roofline > gpurun_out/bench_big$big.log 2>&1
  echo "bench big=$big: $(tail -1 gpurun_out/bench_big$big.log | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["value"], d["final_loss"])')"
done
timeout -k 10 300 python scripts/parity_depth.py > gpurun_out/parity_depth.log 2>&1; echo "parity depth exit $?"; grep "^L=" gpurun_out/parity_depth.log
