#!/bin/bash
# Round 4: 256 x 256 dW tiles (one workgroup per CU, ring of four 32-deep half steps) against the 128 x 128 form.
set -o pipefail
mkdir -p gpurun_out
export OVQA_NO_BUILD=1
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -q -x -p no:cacheprovider -k "wgrad or bwd_weight" > gpurun_out/dw256_tests.log 2>&1 || { tail -30 gpurun_out/dw256_tests.log; exit 1; }
tail -2 gpurun_out/dw256_tests.log
for rep in 1 2; do
  for v in 0 1; do
    OVQA_DW_TILE256=$v timeout -k 10 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-roofline --repeats 3 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('TILE256', $v, d['ms_per_step'])"
  done
done
[ -n "$SKIP_TRAIN" ] || { timeout -k 10 900 python -m pytest tests/test_train_gpu.py tests/test_blocks_gpu.py -q -x -p no:cacheprovider > gpurun_out/dw256_train.log 2>&1; tail -3 gpurun_out/dw256_train.log; }
