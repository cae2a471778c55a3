#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
# never compile from a profiled run (hipcc under the profiler preload would be an exec after GPU init): build first
python -m openvivqa_amd.build > /dev/null 2>&1 || { echo "library build failed"; exit 1; }
export OVQA_NO_BUILD=1
timeout -k 10 600 python -m pytest tests -q -m gpu -p no:cacheprovider > gpurun_out/tests.log 2>&1
echo "gpu tests exit $?"; grep -E "^(FAILED|ERROR)" gpurun_out/tests.log | head -30; tail -2 gpurun_out/tests.log
timeout -k 10 200 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; echo "smoke exit $?"; tail -3 gpurun_out/smoke.log
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > gpurun_out/bench.log 2>&1; echo "bench exit $?"; tail -1 gpurun_out/bench.log | cut -c1-1200
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-roofline > $GRAFT_REPO_ROOT/gpurun_out/prof_bench.log 2>&1
echo "rocprof exit $?"
find $GRAFT_REPO_ROOT/gpurun_out/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} sh -c 'head -24 {}' | cut -c1-150
