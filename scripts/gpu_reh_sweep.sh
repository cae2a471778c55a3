#!/bin/bash
for ov in 0 96 64 32; do
python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-roofline --rehearse-comm --overlap-mb $ov 2>/dev/null | python -c "
import sys, json
r=json.loads(sys.stdin.read()); print('overlap_mb', $ov, 'segments', r['config']['grad_segments'], 'ms', r['ms_per_step'])"
done
python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-roofline --rehearse-comm --overlap-mb 96 --comm-dtype fp32 2>/dev/null | python -c "
import sys, json
r=json.loads(sys.stdin.read()); print('fp32 comm, overlap 96: segments', r['config']['grad_segments'], 'ms', r['ms_per_step'])"
