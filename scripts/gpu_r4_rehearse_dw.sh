#!/bin/bash
# single-rank RCCL rehearsal (5 gradient segments = 5 grouped dW launches) with 128 x 128 against 256 x 256 dW tiles
set -o pipefail
export OVQA_NO_BUILD=1
for rep in 1 2; do for v in 0 1; do
OVQA_DW_TILE256=$v timeout -k 10 400 python bench.py --rehearse-comm --steps 50 --warmup 10 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('REHEARSE tile256=$v', r['ms_per_step'], r['config']['grad_segments'])"
done; done
