#!/bin/bash
# round 5: fc_o dX inside the 100 x 100 self-attention backward (attn_bwd_roles_mfma_kernel<4,4,true,NBUF>): tests, then the
# step with the form off / ring of 3 / ring of 2
export OVQA_NO_BUILD=1 PYTHONDONTWRITEBYTECODE=1
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py tests/test_blocks_gpu.py -x -q -p no:cacheprovider -k "attention or attn or block or mha" 2>&1 | tail -3 || exit 1
for v in 0 3 2 0 3 2; do
  OVQA_DOBWD_ROLES=$v timeout -k 10 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-roofline --no-secondary --repeats 3 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('STEP roles=$v', d['ms_per_step'], d.get('ms_per_step_median'))" || exit 1
done
