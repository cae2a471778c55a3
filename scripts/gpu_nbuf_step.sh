#!/bin/bash
mkdir -p gpurun_out
for cfg in "2 2" "3 3" "4 4" "2 4" "4 2" "3 4"; do
  set -- $cfg
  echo "=== small NBUF=$1 tiny NBUF=$2"
  OVQA_GEMM_SMALL_NBUF=$1 OVQA_GEMM_TINY_NBUF=$2 python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "
import sys, json
r=json.loads(sys.stdin.read()); print(r['ms_per_step'], r['value'])"
done
