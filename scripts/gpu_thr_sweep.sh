#!/bin/bash
export PYTHONDONTWRITEBYTECODE=1
for t in 330 450 1300; do
  echo "== tiny-tile threshold $t"; OVQA_GEMM_SMALL_TILES=100000 OVQA_GEMM_TINY_TILES=$t timeout -k 10 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline 2>&1 | tail -1 | cut -c1-200
done
echo "== small 320 tiny 450";  OVQA_GEMM_TINY_TILES=450 timeout -k 10 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline 2>&1 | tail -1 | cut -c1-200
