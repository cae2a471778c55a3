#!/bin/bash
export PYTHONDONTWRITEBYTECODE=1
for t in 224 700 1300 1000000; do
  echo "== FLUSH_TILES=$t"; OVQA_WGRAD_FLUSH_TILES=$t timeout -k 10 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline 2>&1 | tail -1 | cut -c1-200
done
