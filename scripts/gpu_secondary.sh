#!/bin/bash
# secondary workloads of bench.py (diagnostics, not the metric): whole MCAN model, M4C greedy decode
set -o pipefail
mkdir -p gpurun_out
python -m openvivqa_amd.build > /dev/null 2>&1 || { echo "library build failed"; exit 1; }
export PYTHONDONTWRITEBYTECODE=1 OVQA_NO_BUILD=1
timeout -k 10 300 python bench.py --workload m4c_decode --steps 5 --warmup 1 > gpurun_out/bench_m4c.log 2>&1; echo "m4c exit $?"; tail -1 gpurun_out/bench_m4c.log | cut -c1-700
timeout -k 10 300 python bench.py --workload model --steps 50 --warmup 10 --repeats 3 > gpurun_out/bench_model.log 2>&1; echo "model exit $?"; tail -1 gpurun_out/bench_model.log | cut -c1-700
