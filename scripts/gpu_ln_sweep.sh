#!/bin/bash
set -e
python -m pytest tests/test_kernels_gpu.py tests/test_blocks_gpu.py -x -q 2>&1 | tail -3
for nb in 0 256 512 1024; do
  echo "=== OVQA_LN_BWD_BLOCKS=$nb"
  OVQA_LN_BWD_BLOCKS=$nb python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "
import sys, json
r=json.loads(sys.stdin.read()); print(r['ms_per_step'], r['value'])"
done
