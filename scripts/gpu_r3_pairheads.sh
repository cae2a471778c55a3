#!/bin/bash
# two heads per 16-wave workgroup in the fused guided-attention kernels (OVQA_QATT_PAIR): tests + A/B in the step
mkdir -p gpurun_out
python -m openvivqa_amd.build > /dev/null 2>&1 || exit 1
export OVQA_NO_BUILD=1
timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "bwd_do or attention_q_fwd" > gpurun_out/pair_tests.log 2>&1; rc=$?; echo "kernel tests exit $rc"; tail -3 gpurun_out/pair_tests.log
[ $rc -eq 0 ] || { grep -E "^E " gpurun_out/pair_tests.log | head -20; exit 1; }
timeout -k 10 900 python -m pytest tests/test_blocks_gpu.py tests/test_modules_gpu.py -x -q -m gpu > gpurun_out/pair_tests2.log 2>&1; rc=$?; echo "block/module tests exit $rc"; tail -3 gpurun_out/pair_tests2.log
[ $rc -eq 0 ] || { grep -E "^E " gpurun_out/pair_tests2.log | head -20; exit 1; }
for p in 0 1 0 1; do
  OVQA_QATT_PAIR=$p timeout -k 10 200 python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-roofline --repeats 3 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('pair=$p ms/step', d['ms_per_step'], d['ms_per_step_median'])"
done
