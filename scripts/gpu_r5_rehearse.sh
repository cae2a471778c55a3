#!/bin/bash
# round 5: what the N > 1 machinery costs on ONE rank (single-rank RCCL group: the all-reduce is a copy, so this prices the
# machinery, not xGMI): plain step, 1 segment, the default segmentation (fp32 and bf16 exchange) -- JSON lines under
# gpurun_out/ (the final ones are copied to profiles/r05_rehearse_*.json) -- and per-kernel replay-window statistics of
# the plain and the default-segmentation step, for the difference.
set -o pipefail
export OVQA_NO_BUILD=1 PYTHONDONTWRITEBYTECODE=1
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05_rehearse
rm -rf $O; mkdir -p $O
B="python3 $R/bench.py --no-cpu-baseline --no-roofline --no-secondary --steps 100 --warmup 10 --repeats 3"
$B > $O/plain.json 2> $O/plain.err || { tail -3 $O/plain.err; exit 1; }
$B --rehearse-comm --overlap-mb 0 > $O/seg1.json 2> $O/seg1.err || { tail -3 $O/seg1.err; exit 1; }
$B --rehearse-comm > $O/seg5.json 2> $O/seg5.err || { tail -3 $O/seg5.err; exit 1; }
$B --rehearse-comm --comm-dtype bf16 > $O/seg5_bf16.json 2> $O/seg5_bf16.err || { tail -3 $O/seg5_bf16.err; exit 1; }
for f in plain seg1 seg5 seg5_bf16; do python3 -c "
import json,sys
d=json.load(open('$O/$f.json')); print('$f', d['ms_per_step'], d['ms_per_step_median'], d['config']['grad_segments'])"; done
if [ "$1" = "trace" ]; then
  cd /tmp && export TMPDIR=/tmp
  for f in plain seg5; do
    extra=""; [ $f = seg5 ] && extra="--rehearse-comm"
    timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace_$f -- python3 $R/bench.py --no-cpu-baseline --no-roofline --no-secondary --repeats 1 --steps 10 --warmup 2 $extra > $O/trace_$f.log 2>&1
    find $O/trace_$f -name "*kernel_trace.csv" | head -1 | xargs -I{} python3 $R/scripts/replay_window_stats.py {} 10 $O/${f}_kernel_stats.csv
    rm -rf $O/trace_$f
  done
fi
