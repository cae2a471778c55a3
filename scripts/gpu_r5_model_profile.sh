#!/bin/bash
# round 5: rocprofv3 --kernel-trace --stats of `bench.py --workload model` (the whole configs/mcan.yaml MCAN through
# build_model) -> profiles/<TAG>_model_kernel_stats.csv: where the model-level step spends its time outside the stacks.
# The stats are taken from the REPLAY window only (the last N graph replays in the kernel trace), with a `per_step`
# column, so nobody has to divide by the number of passes.
set -o pipefail
TAG=${1:-r05}
W=${2:-model}
python -m openvivqa_amd.build > /dev/null 2>&1 || { echo "library build failed"; exit 1; }
export OVQA_NO_BUILD=1 PYTHONDONTWRITEBYTECODE=1
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_${TAG}_$W
rm -rf $OUT; mkdir -p $OUT
python3 $R/bench.py --workload $W --steps 50 --warmup 5 --no-cpu-baseline --no-roofline 2> $OUT/bench.err | cut -c1-600 | tee $OUT/bench.json
cd /tmp && export TMPDIR=/tmp
STEPS=10
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --workload $W --no-cpu-baseline --no-roofline --repeats 1 --steps $STEPS --warmup 2 > $OUT/trace.log 2>&1; echo "trace exit $?"
find $OUT/trace -name "*kernel_trace.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_trace.csv
python3 $R/scripts/replay_window_stats.py $OUT/kernel_trace.csv $STEPS $OUT/${W}_kernel_stats.csv
gzip -f $OUT/kernel_trace.csv
head -40 $OUT/${W}_kernel_stats.csv | cut -c1-200
