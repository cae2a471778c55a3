#!/bin/bash
# per-kernel statistics of the replayed step of one bench.py workload: bash scripts/gpu_prof_workload.sh <workload> <tag>
set -e
W=${1:-decoder_train}; TAG=${2:-r06_$W}
mkdir -p gpurun_out/prof_$TAG
export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_$TAG -o trace -- python3 bench.py --workload $W --steps 10 --warmup 2 \
  --no-cpu-baseline --no-roofline --no-secondary --repeats 1 > gpurun_out/prof_$TAG/bench.json 2> gpurun_out/prof_$TAG/bench.err
CSV=$(find gpurun_out/prof_$TAG -name "*kernel_trace.csv" | head -1)
python3 scripts/replay_window_stats.py "$CSV" 10 gpurun_out/${TAG}_kernel_stats.csv
head -40 gpurun_out/${TAG}_kernel_stats.csv
rm -rf gpurun_out/prof_$TAG/*/ 2>/dev/null || true
