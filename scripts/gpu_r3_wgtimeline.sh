#!/bin/bash
# per-workgroup timeline of the step's GEMM shapes (development probe build; the product library is rebuilt afterwards)
# OVQA_WG_PLACEMENT=1 also prints which CU the k-th workgroup of an XCD lands on
mkdir -p gpurun_out
timeout -k 10 400 python scripts/gemm_wg_timeline.py > gpurun_out/wg_timeline.log 2>&1
echo "timeline exit $?"
grep -v "amdgpu.ids" gpurun_out/wg_timeline.log
python -c "
from openvivqa_amd import build as B; B.build(force=True, verbose=False)"
