#!/bin/bash
for e in 0 1; do
OVQA_DX_WT_EXPERIMENT=$e python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "
import sys, json
r=json.loads(sys.stdin.read()); print('WT experiment', $e, 'ms', r['ms_per_step'])"
done
