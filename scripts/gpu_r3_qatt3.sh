#!/bin/bash
mkdir -p gpurun_out
python -m openvivqa_amd.build > /dev/null 2>&1 || exit 1
export OVQA_NO_BUILD=1
timeout -k 10 900 python -m pytest tests/test_blocks_gpu.py tests/test_modules_gpu.py -x -q -m gpu > gpurun_out/qatt_tests2.log 2>&1; rc=$?; echo "block/module tests exit $rc"; tail -3 gpurun_out/qatt_tests2.log
[ $rc -eq 0 ] || { grep -E "^E " gpurun_out/qatt_tests2.log | head -20; exit 1; }
for p in 1 0 1 0; do
  OVQA_NO_FUSED_Q=$p timeout -k 10 200 python bench.py --workload cross_modality --steps 50 --warmup 10 --repeats 3 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('xmod no_fused_q=$p ms/step', d['ms_per_step'], d['ms_per_step_median'])"
done
