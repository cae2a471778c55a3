#!/bin/bash
mkdir -p gpurun_out
python -m openvivqa_amd.build > /dev/null 2>&1 || exit 1
export OVQA_NO_BUILD=1
timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "attention_q_fwd" > gpurun_out/qatt_tests.log 2>&1; rc=$?; echo "kernel tests exit $rc"; tail -2 gpurun_out/qatt_tests.log
for p in 2 3 2 3; do
  OVQA_QATT_NBUF=$p timeout -k 10 200 python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-roofline --repeats 3 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('nbuf=$p ms/step', d['ms_per_step'], d['ms_per_step_median'])"
done
