#!/bin/bash
# Fused projection + attention forward: kernel test first (small, under its own timeout), then the attention / module
# suites, then the step with and without the fused kernel, then the default bench (in-step roofline calibration).
set -o pipefail
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py -q -m gpu -p no:cacheprovider -k "attention_qkv_fwd" > gpurun_out/fused_test.log 2>&1
rc=$?; echo "fused kernel test exit $rc"; tail -5 gpurun_out/fused_test.log
[ $rc -eq 0 ] || { grep -E "^(FAILED|ERROR|E )" gpurun_out/fused_test.log | head -30; exit 1; }
timeout -k 10 900 python -m pytest tests -q -m gpu -p no:cacheprovider > gpurun_out/tests.log 2>&1
echo "gpu tests exit $?"; grep -E "^(FAILED|ERROR)" gpurun_out/tests.log | head -40; tail -3 gpurun_out/tests.log
timeout -k 10 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-roofline --repeats 3 > gpurun_out/bench_fused.log 2>&1; echo "bench (fused) exit $?"; tail -1 gpurun_out/bench_fused.log | cut -c1-300
OVQA_NO_FUSED_QKV=1 timeout -k 10 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-roofline --repeats 3 > gpurun_out/bench_unfused.log 2>&1; echo "bench (separate) exit $?"; tail -1 gpurun_out/bench_unfused.log | cut -c1-300
timeout -k 10 400 python bench.py > gpurun_out/bench_default.log 2>&1; echo "bench default exit $?"; tail -1 gpurun_out/bench_default.log | cut -c1-1500
