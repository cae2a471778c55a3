#!/bin/bash
# A/B on one box: the committed library (HEAD) against the working tree's, alternated; kernel + module tests first
set -o pipefail
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
python -m openvivqa_amd.build > /dev/null 2>&1 || { echo "library build failed"; exit 1; }
timeout -k 10 900 python -m pytest tests/test_kernels_gpu.py tests/test_modules_gpu.py -q -m gpu -p no:cacheprovider -x > gpurun_out/km_test.log 2>&1
rc=$?; echo "kernel+module tests exit $rc"; tail -3 gpurun_out/km_test.log
[ $rc -eq 0 ] || { grep -E "^(FAILED|ERROR|E )" gpurun_out/km_test.log | head -30; exit 1; }
cp openvivqa_amd/csrc/libovqa_hip.so gpurun_out/lib_new.so
[ -f scripts/lib_base.so ] || { echo "no baseline library (scripts/lib_base.so)"; exit 1; }
bench() {
  cp $2 openvivqa_amd/csrc/libovqa_hip.so
  OVQA_NO_BUILD=1 timeout -k 10 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-roofline --repeats 3 2>/dev/null | python -c "
import sys, json
r = json.loads(sys.stdin.read()); print('$1', r['ms_per_step'], r['ms_per_step_min'], r['ms_per_step_max'], r['final_loss'])"
}
for i in 1 2 3; do
  bench base scripts/lib_base.so
  bench new gpurun_out/lib_new.so
done
cp gpurun_out/lib_new.so openvivqa_amd/csrc/libovqa_hip.so; rm -f gpurun_out/lib_new.so
