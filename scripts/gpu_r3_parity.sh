#!/bin/bash
# Round 3: the -m gpu suite with the parity report (every compared quantity, measured error and bar)
set -o pipefail
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
python -m openvivqa_amd.build > /dev/null 2>&1 || { echo "library build failed"; exit 1; }
export OVQA_NO_BUILD=1
rm -f gpurun_out/parity_report.tsv
OVQA_PARITY_REPORT=$PWD/gpurun_out/parity_report.tsv timeout -k 10 1000 python -m pytest tests -q -m gpu ${PYTEST_K:+-k "$PYTEST_K"} -p no:cacheprovider > gpurun_out/tests.log 2>&1
echo "gpu tests exit $?"; grep -E "^(FAILED|ERROR)" gpurun_out/tests.log | cut -c1-300 | head -60; tail -3 gpurun_out/tests.log
wc -l gpurun_out/parity_report.tsv
