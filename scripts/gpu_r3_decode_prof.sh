#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
python -m openvivqa_amd.build > /dev/null 2>&1 || { echo "library build failed"; exit 1; }
export OVQA_NO_BUILD=1
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/decode_prof
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/decode_prof -- python3 $R/bench.py --workload decode --beam ${PROF_BEAM:-3} --steps 10 --warmup 1 > $R/gpurun_out/decode_prof.log 2>&1
echo "rocprof exit $?"
find $R/gpurun_out/decode_prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $R/gpurun_out/decode_kernel_stats.csv
