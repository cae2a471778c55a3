#!/bin/bash
# per-kernel time of the grouped dW launch, 128 x 128 against 256 x 256 tiles (rocprofv3 --kernel-trace --stats)
set -o pipefail
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for v in ${VARIANTS:-0 1}; do
  export OVQA_DW_TILE256=$v OVQA_NO_BUILD=1
  rm -rf /tmp/prof_$v
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$v -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-roofline --repeats 1 > $R/gpurun_out/dwprof_$v.log 2>&1
  echo "== TILE256=$v exit $?"
  f=$(find /tmp/prof_$v -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && { cp $f $R/gpurun_out/dwprof_${v}_kernel_stats.csv; grep -i "wgrad" $f | cut -c1-60,100-400; }
done
