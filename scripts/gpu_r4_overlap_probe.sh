#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
timeout -k 10 500 python scripts/overlap_probe.py > gpurun_out/overlap_probe.json 2> gpurun_out/overlap_probe.err || { echo probe failed; tail -20 gpurun_out/overlap_probe.err; exit 1; }
python -c "
import json; r=json.loads(open('gpurun_out/overlap_probe.json').read().strip().splitlines()[-1])
m=r.pop('masked'); print(r)
for k,v in m.items(): print(k, v)"
