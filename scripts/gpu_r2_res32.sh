#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
timeout -k 10 200 python scripts/floor_probe.py > gpurun_out/floor_probe.log 2>&1; echo "floor probe exit $?"; tail -14 gpurun_out/floor_probe.log
timeout -k 10 300 python scripts/parity_l6.py 16 > gpurun_out/parity_l6.log 2>&1; echo "parity exit $?"; tail -5 gpurun_out/parity_l6.log
timeout -k 10 900 python -m pytest tests -q -m gpu -p no:cacheprovider > gpurun_out/tests.log 2>&1
echo "gpu tests exit $?"; grep -E "^(FAILED|ERROR)" gpurun_out/tests.log | head -40; tail -3 gpurun_out/tests.log
timeout -k 10 300 python bench.py --steps 50 --warmup 10 --no-cpu-baseline > gpurun_out/bench_res32.log 2>&1; echo "bench exit $?"; tail -1 gpurun_out/bench_res32.log | cut -c1-400
