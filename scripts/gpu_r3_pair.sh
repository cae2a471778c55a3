#!/bin/bash
# co-resident workgroups on neighbouring tiles (OVQA_GEMM_PAIR): per-workgroup timelines, kernel tests, step
mkdir -p gpurun_out
OVQA_GEMM_PAIR=0 timeout -k 10 300 python scripts/gemm_wg_timeline.py > gpurun_out/wg_pair0.log 2>&1 || { tail -5 gpurun_out/wg_pair0.log; exit 1; }
OVQA_PROBE_BUILD=0 OVQA_GEMM_PAIR=1 timeout -k 10 300 python scripts/gemm_wg_timeline.py > gpurun_out/wg_pair1.log 2>&1 || { tail -5 gpurun_out/wg_pair1.log; exit 1; }
for f in wg_pair0 wg_pair1; do echo "#### $f"; grep "^==\|span\|K loop\|CUs holding" gpurun_out/$f.log | cut -c1-160; done
python -c "
from openvivqa_amd import build as B; B.build(force=True, verbose=False)" || exit 1
export OVQA_NO_BUILD=1
OVQA_GEMM_PAIR=1 timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "linear or gemm" > gpurun_out/pair_tests.log 2>&1; echo "tests exit $?"; tail -2 gpurun_out/pair_tests.log
for p in 0 1 0 1; do
  OVQA_GEMM_PAIR=$p timeout -k 10 200 python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-roofline --repeats 3 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('pair=$p ms/step', d['ms_per_step'], d['ms_per_step_median'])"
done
