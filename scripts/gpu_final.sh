#!/bin/bash
# end-of-round check: GPU tests, smoke, the default bench line, and the step profiles for profiles/
set -o pipefail
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
TAG=${1:-r02}
timeout -k 10 900 python -m pytest tests -q -m gpu -p no:cacheprovider > gpurun_out/tests.log 2>&1
echo "gpu tests exit $?"; grep -E "^(FAILED|ERROR)" gpurun_out/tests.log | head -40; tail -3 gpurun_out/tests.log
timeout -k 10 200 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; echo "smoke exit $?"; tail -3 gpurun_out/smoke.log
timeout -k 10 600 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; echo "bench exit $?"; tail -1 gpurun_out/bench_default.json | cut -c1-4000
bash scripts/gpu_profile_step.sh $TAG
