#!/bin/bash
# segment-count sweep of the single-rank rehearsal (see gpu_r5_rehearse.sh): --overlap-mb -> segments, ms per step
export OVQA_NO_BUILD=1 PYTHONDONTWRITEBYTECODE=1
R=$GRAFT_REPO_ROOT
B="python3 $R/bench.py --no-cpu-baseline --no-roofline --no-secondary --steps 100 --warmup 10 --repeats 3"
$B 2>/dev/null | python3 -c "import json,sys; d=json.load(sys.stdin); print('plain', d['ms_per_step'], d['ms_per_step_median'])"
for mb in 0 32 48 64 80 96 128; do
  $B --rehearse-comm --overlap-mb $mb 2>/dev/null | python3 -c "import json,sys; d=json.load(sys.stdin); print('overlap-mb $mb segments', d['config']['grad_segments'], d['ms_per_step'], d['ms_per_step_median'])"
done
