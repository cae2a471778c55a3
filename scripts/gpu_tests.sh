#!/bin/bash
# the -m gpu suite + smoke, logs under gpurun_out/
set -o pipefail
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
timeout -k 10 900 python -m pytest tests -q -m gpu -p no:cacheprovider > gpurun_out/tests.log 2>&1
echo "gpu tests exit $?"; grep -E "^(FAILED|ERROR)" gpurun_out/tests.log | head -40; tail -3 gpurun_out/tests.log
timeout -k 10 200 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; echo "smoke exit $?"; tail -3 gpurun_out/smoke.log
