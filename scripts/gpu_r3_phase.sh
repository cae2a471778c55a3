#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 400 python scripts/phase_probe.py > gpurun_out/phase_probe.log 2>&1; echo "probe exit $?"
grep "fused fwd" gpurun_out/phase_probe.log | cut -c1-220
python -c "
from openvivqa_amd import build as B; B.build(force=True, verbose=False)"
