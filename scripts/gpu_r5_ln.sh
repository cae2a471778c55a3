#!/bin/bash
# round 5: LayerNorm forms alone (scripts/ln_bench.py: cold and warm, one process per form), then the step with the candidates
export OVQA_NO_BUILD=1 PYTHONDONTWRITEBYTECODE=1
mkdir -p gpurun_out
: > gpurun_out/ln_bench.jsonl
for w in 8 13 16; do
  OVQA_LN_BWD_WAVES=$w timeout -k 10 120 python scripts/ln_bench.py 6400 >> gpurun_out/ln_bench.jsonl || exit 1
done
cat gpurun_out/ln_bench.jsonl
for w in 8 13 16 8 13; do
  OVQA_LN_BWD_WAVES=$w timeout -k 10 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-roofline --no-secondary --repeats 3 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('STEP waves=$w', d['ms_per_step'], d.get('ms_per_step_median'))" || exit 1
done
