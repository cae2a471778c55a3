#!/bin/bash
mkdir -p gpurun_out
for nb in 2 3 4; do
  echo "=== NBUF small=tiny=$nb" 
  OVQA_GEMM_SMALL_NBUF=$nb OVQA_GEMM_TINY_NBUF=$nb python scripts/gemm_bench.py fwd bwd_data 2>&1 | grep -v amdgpu | python -c "
import sys, json
for l in sys.stdin:
    try: r=json.loads(l)
    except Exception: print(l.strip()); continue
    print(r['M'],r['N'],r['K'],'bias',r.get('bias'),'gelu',r.get('gelu'),'resid',r.get('resid'),'dX',r.get('dX'))
"
done
