#!/bin/bash
# per-workgroup timeline of the step's GEMM shapes (development probe build; the product library is rebuilt afterwards)
mkdir -p gpurun_out
OVQA_GEMM_QREG=0 timeout -k 10 300 python scripts/gemm_wg_timeline.py > gpurun_out/wg_timeline.log 2>&1 || exit 1
for q in 1 2 3; do
  OVQA_PROBE_BUILD=0 OVQA_GEMM_QREG=$q timeout -k 10 200 python scripts/gemm_wg_timeline.py >> gpurun_out/wg_timeline.log 2>&1 || exit 1
done
grep -v "^   start\|^   end\|workgroups that started\|amdgpu.ids" gpurun_out/wg_timeline.log
python -c "
from openvivqa_amd import build as B; B.build(force=True, verbose=False)" && timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "linear or gemm or wgrad" > gpurun_out/qreg_tests.log 2>&1
echo "tests exit $?"; tail -5 gpurun_out/qreg_tests.log
