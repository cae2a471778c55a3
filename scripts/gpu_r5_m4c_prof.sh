#!/bin/bash
# round 5: where a 12-pass M4C decode spends its time (rocprofv3 --kernel-trace --stats of bench.py --workload m4c_decode)
export OVQA_NO_BUILD=1 PYTHONDONTWRITEBYTECODE=1
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_m4c
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --workload m4c_decode --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-secondary > $OUT/trace.log 2>&1; echo "exit $?"
f=$(find $OUT/trace -name "*kernel_stats.csv" | head -1)
cp $f $OUT/kernel_stats.csv
head -25 $OUT/kernel_stats.csv | cut -c1-100,101-260
rm -rf $OUT/trace
