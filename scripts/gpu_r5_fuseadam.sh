#!/bin/bash
# round 5: Adam inside the last grouped weight-gradient launch (TrainStep(fuse_adam=True), world size 1): tests, then the step
# with and without it (time and per-kernel statistics)
export OVQA_NO_BUILD=1 PYTHONDONTWRITEBYTECODE=1
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_train_gpu.py tests/test_kernels_gpu.py -x -q -p no:cacheprovider -k "adam or train or step or determin or wgrad or grouped" 2>&1 | tail -3 || exit 1
for extra in "" "--no-fuse-adam" "" "--no-fuse-adam"; do
  timeout -k 10 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-roofline --no-secondary --repeats 3 $extra 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('STEP [$extra]', d['ms_per_step'], d.get('ms_per_step_median'), d['config']['adam'][:40])" || exit 1
done
