#!/bin/bash
mkdir -p gpurun_out
python -c "
from openvivqa_amd import build as B; B.build(verbose=False)" || exit 1
export OVQA_NO_BUILD=1
for p in 0 3 4 0 3 4; do
  OVQA_GEMM_BIG16=$p timeout -k 10 200 python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-roofline --repeats 3 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('big16=$p ms/step', d['ms_per_step'], d['ms_per_step_median'])"
done
