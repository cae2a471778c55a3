#!/bin/bash
# round 5: the LSTM kernels first (short timeout: a persistent launch that hangs must not take the box), then the module level
set -o pipefail
export OVQA_NO_BUILD=1 PYTHONDONTWRITEBYTECODE=1
mkdir -p gpurun_out
timeout -k 10 400 python -m pytest tests/test_kernels_gpu.py -x -q -k "lstm" 2>&1 | tail -25 || exit 1
timeout -k 10 400 python -m pytest tests/test_modules_gpu.py -x -q -k "G17 or G12" 2>&1 | tail -15 || exit 1
timeout -k 10 300 python bench.py --workload model --steps 30 --warmup 5 --no-cpu-baseline --no-roofline 2> gpurun_out/bench_model.err | cut -c1-300 || { tail -5 gpurun_out/bench_model.err; exit 1; }
