#!/bin/bash
# epilogue operand prefetch: timelines, GEMM tests, step
mkdir -p gpurun_out
timeout -k 10 300 python scripts/gemm_wg_timeline.py > gpurun_out/wg_epi.log 2>&1 || { tail -5 gpurun_out/wg_epi.log; exit 1; }
grep "^==\|span\|K loop" gpurun_out/wg_epi.log | cut -c1-150
python -c "
from openvivqa_amd import build as B; B.build(force=True, verbose=False)" || exit 1
export OVQA_NO_BUILD=1
timeout -k 10 900 python -m pytest tests/test_kernels_gpu.py tests/test_blocks_gpu.py -x -q -m gpu > gpurun_out/epi_tests.log 2>&1; echo "tests exit $?"; tail -2 gpurun_out/epi_tests.log
for p in 1 2 3; do
  timeout -k 10 200 python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-roofline --repeats 3 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ms/step', d['ms_per_step'], d['ms_per_step_median'])"
done
