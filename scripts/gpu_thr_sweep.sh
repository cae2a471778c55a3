#!/bin/bash
# In-step sweep of the GEMM tile-tier thresholds and ring depths (bench.py ms/step for each setting).
export PYTHONDONTWRITEBYTECODE=1
run() { env "$@" python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "
import sys, json
r=json.loads(sys.stdin.read()); print('$*', r['ms_per_step'])"; }
run A=0
run OVQA_GEMM_SMALL_TILES=640
run OVQA_GEMM_SMALL_TILES=900
run OVQA_GEMM_SMALL_TILES=640 OVQA_GEMM_TINY_TILES=450
run OVQA_GEMM_TINY_TILES=450
run OVQA_GEMM_TINY_TILES=200
run OVQA_GEMM_SMALL_NBUF=2
run OVQA_GEMM_TINY_NBUF=3
run OVQA_GEMM_TINY_NBUF=2 OVQA_GEMM_SMALL_NBUF=2
run OVQA_GEMM_VARIANT=13
