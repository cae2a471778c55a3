#!/bin/bash
# the 128 x 128 dW form with untracked LDS reads: k-split wave grids on / off, in the step and on its own
set -o pipefail
export OVQA_NO_BUILD=1 OVQA_DW_TILE256=0
for ks in 0 1; do OVQA_DW_KSPLIT=$ks timeout -k 10 200 python scripts/dw_bench.py 2>&1 | tail -1; done
for rep in 1 2; do for ks in 0 1; do
OVQA_DW_KSPLIT=$ks timeout -k 10 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-roofline --repeats 3 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('KSPLIT', $ks, d['ms_per_step'])"
done; done
