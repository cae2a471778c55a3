#!/bin/bash
# Round 3: decode workload (BASELINE configs[4]) -- timing and kernel profile
set -o pipefail
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
python -m openvivqa_amd.build > /dev/null 2>&1 || { echo "library build failed"; exit 1; }
export OVQA_NO_BUILD=1
for beam in 1 3; do
  timeout -k 10 300 python bench.py --workload decode --beam $beam --steps 10 --warmup 2 2> gpurun_out/decode_b$beam.err | cut -c1-1200
done
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/decode_prof
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/decode_prof -- python3 $R/bench.py --workload decode --beam ${PROF_BEAM:-3} --steps 3 --warmup 1 > $R/gpurun_out/decode_prof.log 2>&1
echo "rocprof exit $?"
find $R/gpurun_out/decode_prof -name "*kernel_stats.csv" | head -1 | xargs -I{} sh -c 'cp {} '$R'/gpurun_out/decode_kernel_stats.csv; head -28 {}' | cut -c1-170
