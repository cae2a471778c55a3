#!/bin/bash
# one 16-wave workgroup on a 128 x 128 tile instead of two 8-wave ones on 128 x 64 tiles (OVQA_GEMM_BIG16)
mkdir -p gpurun_out
OVQA_GEMM_BIG16=0 timeout -k 10 300 python scripts/gemm_wg_timeline.py > gpurun_out/wg_b0.log 2>&1 || { tail -5 gpurun_out/wg_b0.log; exit 1; }
OVQA_PROBE_BUILD=0 OVQA_GEMM_BIG16=3 timeout -k 10 300 python scripts/gemm_wg_timeline.py > gpurun_out/wg_b3.log 2>&1 || { tail -5 gpurun_out/wg_b3.log; exit 1; }
OVQA_PROBE_BUILD=0 OVQA_GEMM_BIG16=2 timeout -k 10 300 python scripts/gemm_wg_timeline.py > gpurun_out/wg_b2.log 2>&1 || { tail -5 gpurun_out/wg_b2.log; exit 1; }
for f in wg_b0 wg_b3 wg_b2; do echo "#### $f"; grep "^==\|span\|K loop\|CUs holding" gpurun_out/$f.log | grep -A4 "x512 <-" | cut -c1-160; done
python -c "
from openvivqa_amd import build as B; B.build(force=True, verbose=False)" || exit 1
export OVQA_NO_BUILD=1
OVQA_GEMM_BIG16=3 timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "linear or gemm" > gpurun_out/big16_tests.log 2>&1; echo "tests exit $?"; tail -2 gpurun_out/big16_tests.log
for p in 0 3 2 0 3; do
  OVQA_GEMM_BIG16=$p timeout -k 10 200 python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-roofline --repeats 3 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('big16=$p ms/step', d['ms_per_step'], d['ms_per_step_median'])"
done
