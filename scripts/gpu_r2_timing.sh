#!/bin/bash
# launch-timing hook test + the default bench line (in-step roofline through hipExtLaunchKernel events)
set -o pipefail
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py -q -m gpu -p no:cacheprovider -k "launch_timing or dispatch_hook or attention_qkv" > gpurun_out/timing_test.log 2>&1
rc=$?; echo "timing test exit $rc"; tail -3 gpurun_out/timing_test.log
[ $rc -eq 0 ] || { grep -E "^(FAILED|ERROR|E )" gpurun_out/timing_test.log | head -30; exit 1; }
timeout -k 10 600 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; echo "bench exit $?"; tail -1 gpurun_out/bench_default.json | cut -c1-2600
