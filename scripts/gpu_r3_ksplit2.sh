#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
OVQA_GEMM_KSPLIT=0 timeout -k 10 300 python scripts/gemm_phase_probe.py 2>&1 | grep ksplit
OVQA_PROBE_BUILD=0 OVQA_GEMM_KSPLIT=3 timeout -k 10 300 python scripts/gemm_phase_probe.py 2>&1 | grep ksplit
# back to the production build
python -m openvivqa_amd.build --force > /dev/null 2>&1 || { echo "library build failed"; exit 1; }
export OVQA_NO_BUILD=1
for ks in 0 3; do
  echo "== gemm_bench ksplit=$ks"
  OVQA_GEMM_KSPLIT=$ks timeout -k 10 300 python scripts/gemm_bench.py fwd bwd_data 2>&1 | grep "6400\|8192" | cut -c1-400
done
