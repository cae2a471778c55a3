#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py tests/test_train_gpu.py tests/test_blocks_gpu.py -q -m gpu -p no:cacheprovider -x > gpurun_out/tests_epi.log 2>&1
echo "tests exit $?"; grep -E "^(FAILED|ERROR)" gpurun_out/tests_epi.log | head; tail -2 gpurun_out/tests_epi.log
timeout -k 10 200 python scripts/gemm_bench.py fwd bwd_data > gpurun_out/gemm_bench_epi.log 2>&1; echo "gemm bench exit $?"
cut -c1-300 gpurun_out/gemm_bench_epi.log | grep -v amdgpu.ids | head -9
timeout -k 10 600 python bench.py --steps 100 --warmup 10 > gpurun_out/bench_full.log 2>gpurun_out/bench_full.err; echo "bench exit $?"; tail -1 gpurun_out/bench_full.log | cut -c1-3000; tail -3 gpurun_out/bench_full.err
