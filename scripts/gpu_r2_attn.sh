#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
timeout -k 10 900 python -m pytest tests -q -m gpu -p no:cacheprovider > gpurun_out/tests.log 2>&1
echo "gpu tests exit $?"; grep -E "^(FAILED|ERROR)" gpurun_out/tests.log | head -40; tail -3 gpurun_out/tests.log
timeout -k 10 300 python bench.py --workload m4c_decode --steps 5 --warmup 1 > gpurun_out/bench_m4c.log 2>&1; echo "m4c bench exit $?"; tail -1 gpurun_out/bench_m4c.log | cut -c1-900
OVQA_FORCE_SIMPLE=1 timeout -k 10 300 python bench.py --workload m4c_decode --steps 2 --warmup 1 > gpurun_out/bench_m4c_simple.log 2>&1; echo "m4c bench (VALU kernels) exit $?"; tail -1 gpurun_out/bench_m4c_simple.log | cut -c1-300
timeout -k 10 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-roofline --repeats 3 > gpurun_out/bench_now.log 2>&1; echo "bench exit $?"; tail -1 gpurun_out/bench_now.log | cut -c1-400
