#!/bin/bash
mkdir -p gpurun_out
OVQA_WG_PLACEMENT=1 timeout -k 10 300 python scripts/gemm_wg_timeline.py > gpurun_out/wg_place.log 2>&1 || { tail -5 gpurun_out/wg_place.log; exit 1; }
grep -A2 "^==\|blockIdx %" gpurun_out/wg_place.log | grep "^==\|blockIdx %" | cut -c1-700
python -c "
from openvivqa_amd import build as B; B.build(force=True, verbose=False)"
