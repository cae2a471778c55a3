#!/bin/bash
# round 5: per-kernel statistics of the replayed CrossModalityTransformer step (bench.py --workload cross_modality)
export OVQA_NO_BUILD=1 PYTHONDONTWRITEBYTECODE=1
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_xmod
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 $R/bench.py --workload cross_modality --steps 10 --warmup 2 --no-cpu-baseline --no-roofline --no-secondary --repeats 1 > $OUT/trace.log 2>&1; echo "exit $?"
f=$(find $OUT/trace -name "*kernel_trace.csv" | head -1)
python3 $R/scripts/replay_window_stats.py $f 10 $OUT/kernel_stats.csv
rm -rf $OUT/trace
