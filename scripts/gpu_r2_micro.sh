#!/bin/bash
# 64-row four-wave tiles for the products with few tiles: GEMM tests with the tiers forced on, then a sweep of the step
set -o pipefail
mkdir -p gpurun_out
python -m openvivqa_amd.build > /dev/null 2>&1 || { echo "library build failed"; exit 1; }
export PYTHONDONTWRITEBYTECODE=1 OVQA_NO_BUILD=1
OVQA_GEMM_MICRO_TILES=100000 OVQA_GEMM_MICRO64_TILES=100000 timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -q -m gpu -p no:cacheprovider -x -k "linear or dropout or gemm or dispatch" > gpurun_out/micro_test.log 2>&1
rc=$?; echo "GEMM tests (micro tiles forced) exit $rc"; tail -3 gpurun_out/micro_test.log
[ $rc -eq 0 ] || { grep -E "^(FAILED|ERROR|E )" gpurun_out/micro_test.log | head -30; exit 1; }
run() {
  OVQA_GEMM_MICRO_TILES=$1 OVQA_GEMM_MICRO64_TILES=$2 OVQA_GEMM_MICRO_NBUF=$3 timeout -k 10 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-roofline --repeats 3 2>/dev/null | python -c "
import sys, json
r = json.loads(sys.stdin.read()); print('micro32<=$1 micro64<=$2 nbuf=$3', r['ms_per_step'], r['ms_per_step_min'], r['ms_per_step_max'])"
}
for i in 1 2; do
  run 0 0 4
  run 100000 0 4
  run 100000 0 3
  run 100000 0 6
  run 100000 100000 4
  run 100000 100000 3
  run 100000 200 4
done
