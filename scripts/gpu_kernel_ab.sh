#!/bin/bash
# Per-kernel A/B of the replayed training step: one rocprofv3 --kernel-trace run per environment setting, statistics of the
# replay window only (scripts/replay_window_stats.py), the rows matching PATTERN and the step's kernel-time sum.
#   bash scripts/gpu_kernel_ab.sh PATTERN "VAR=1" "VAR=2 OTHER=3" ...      ("-" = no variable)
set -o pipefail
PAT=$1; shift
export OVQA_NO_BUILD=1 PYTHONDONTWRITEBYTECODE=1
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/kernel_ab
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for cfg in "$@"; do
  i=$((i+1))
  (
    [ "$cfg" != "-" ] && export $cfg
    timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/t$i -- python3 $R/bench.py --no-cpu-baseline --no-roofline --repeats 1 --no-secondary --steps 10 --warmup 2 > $OUT/t$i.log 2>&1
  ) || { echo "run $i ($cfg) failed"; tail -5 $OUT/t$i.log; exit 1; }
  f=$(find $OUT/t$i -name "*kernel_trace.csv" | head -1)
  python3 $R/scripts/replay_window_stats.py $f 10 $OUT/stats_$i.csv > /dev/null || exit 1
  ARGS="$ARGS ${cfg// /,}::=$OUT/stats_$i.csv"
  rm -rf $OUT/t$i
done
python3 $R/scripts/kernel_ab_table.py "$PAT" $ARGS
