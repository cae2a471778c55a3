#!/bin/bash
# fc_o dX inside the guided-attention backward kernel: tests, A/B in the step (OVQA_NO_FUSED_DO=1 = two kernels)
mkdir -p gpurun_out
python -m openvivqa_amd.build > /dev/null 2>&1 || exit 1
export OVQA_NO_BUILD=1
timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "bwd_do or attention_bwd or smallk or merged" > gpurun_out/bwddo_tests.log 2>&1; rc=$?; echo "kernel tests exit $rc"; tail -3 gpurun_out/bwddo_tests.log
[ $rc -eq 0 ] || { grep -E "^E " gpurun_out/bwddo_tests.log | head -20; exit 1; }
timeout -k 10 900 python -m pytest tests/test_blocks_gpu.py tests/test_modules_gpu.py -x -q -m gpu > gpurun_out/bwddo_tests2.log 2>&1; rc=$?; echo "block/module tests exit $rc"; tail -3 gpurun_out/bwddo_tests2.log
[ $rc -eq 0 ] || { grep -E "^E " gpurun_out/bwddo_tests2.log | head -20; exit 1; }
for p in 1 0 1 0; do
  OVQA_NO_FUSED_DO=$p timeout -k 10 200 python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-roofline --repeats 3 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('no_fused_do=$p ms/step', d['ms_per_step'], d['ms_per_step_median'], 'loss', d['final_loss'])"
done
