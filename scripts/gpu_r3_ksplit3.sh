#!/bin/bash
# k-split wave grids: correctness under the switches, then step A/B (alternated)
set -o pipefail
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
python -m openvivqa_amd.build > /dev/null 2>&1 || { echo "library build failed"; exit 1; }
export OVQA_NO_BUILD=1
OVQA_GEMM_KSPLIT=3 OVQA_GEMM_KSPLIT_MINK=64 OVQA_DW_KSPLIT=1 timeout -k 10 900 python -m pytest tests/test_kernels_gpu.py tests/test_blocks_gpu.py tests/test_train_gpu.py -q -x -p no:cacheprovider > gpurun_out/ksplit_tests.log 2>&1
echo "tests under ksplit exit $?"; tail -2 gpurun_out/ksplit_tests.log
run() {
  OVQA_GEMM_KSPLIT=$1 OVQA_DW_KSPLIT=$2 timeout -k 10 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-roofline --repeats 3 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); print('gemm_ksplit=$1 dw_ksplit=$2', d['ms_per_step'], d['ms_per_step_min'], d['ms_per_step_max'])"
}
for rep in 1 2; do
  run 0 0; run 3 0; run 0 1; run 3 1; run 1 1; run 2 1
done
