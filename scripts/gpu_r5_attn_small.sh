#!/bin/bash
# round 5: fused-dO backward kernels with their operands requested before the projection loop (tests), then step A/B of
# the ring-depth forms of the two 20 x 20 kernels
export OVQA_NO_BUILD=1 PYTHONDONTWRITEBYTECODE=1
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py tests/test_blocks_gpu.py -x -q -p no:cacheprovider -k "attention or attn or block or mha" 2>&1 | tail -3 || exit 1
step() {
  env "$@" timeout -k 10 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-roofline --no-secondary --repeats 3 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('STEP $*', d['ms_per_step'], d.get('ms_per_step_median'))" || exit 1
}
step OVQA_DOBWD_ROLES=3
step OVQA_DOBWD_ROLES=0
step OVQA_QKV_TEXT_FORM=1
step OVQA_QKV_TEXT_FORM=2
step OVQA_QKV_TEXT_FORM=3
step OVQA_DOBWD1_NBUF=4
step OVQA_DOBWD1_NBUF=5
step OVQA_DOBWD_ROLES=3
step OVQA_DOBWD_ROLES=0
