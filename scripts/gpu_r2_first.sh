#!/bin/bash
# round 2, first GPU call: boundary microbenchmark, baseline bench on this box, full kernel trace of a few steps
set -o pipefail
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
# never compile from a profiled run (hipcc under the profiler preload would be an exec after GPU init): build first
python -m openvivqa_amd.build > /dev/null 2>&1 || { echo "library build failed"; exit 1; }
export OVQA_NO_BUILD=1
R=$GRAFT_REPO_ROOT
timeout -k 10 300 python scripts/boundary_bench.py > gpurun_out/boundary.log 2>&1; echo "boundary exit $?"; tail -40 gpurun_out/boundary.log
timeout -k 10 300 python bench.py --steps 50 --warmup 10 > gpurun_out/bench_r2_base.log 2>&1; echo "bench exit $?"; tail -1 gpurun_out/bench_r2_base.log | cut -c1-600
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/trace_base
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/trace_base -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-roofline > $R/gpurun_out/trace_base.log 2>&1
echo "rocprof exit $?"
find $R/gpurun_out/trace_base -name "*kernel_trace.csv" | head -1 | xargs -I{} sh -c 'wc -l {}; gzip -f {}'
