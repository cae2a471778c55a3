#!/bin/bash
set -e
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q 2>&1 | tee gpurun_out/tests.log | tail -4
python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "
import sys, json
r=json.loads(sys.stdin.read()); print('STEP', r['ms_per_step'], r['value'], r['final_loss'])"
