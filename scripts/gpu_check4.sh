#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
timeout -k 10 600 python -m pytest tests -q -m gpu -p no:cacheprovider > gpurun_out/tests.log 2>&1
echo "gpu tests exit $?"; grep -E "^(FAILED|ERROR)" gpurun_out/tests.log | head -30; tail -2 gpurun_out/tests.log
timeout -k 10 200 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; echo "smoke exit $?"; tail -3 gpurun_out/smoke.log
timeout -k 10 300 python scripts/gemm_bench.py > gpurun_out/gemm_bench.log 2>&1; echo "gemm bench exit $?"; grep -v amdgpu.ids gpurun_out/gemm_bench.log
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > gpurun_out/bench.log 2>&1; echo "bench exit $?"; tail -1 gpurun_out/bench.log | cut -c1-1500
