#!/bin/bash
# never compile from a profiled run (hipcc under the profiler preload would be an exec after GPU init): build first
python -m openvivqa_amd.build > /dev/null 2>&1 || { echo "library build failed"; exit 1; }
export OVQA_NO_BUILD=1
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_reh
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_reh -- python3 $GRAFT_REPO_ROOT/bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-roofline --rehearse-comm > /dev/null 2>&1
ls $GRAFT_REPO_ROOT/gpurun_out/prof_reh/*/
