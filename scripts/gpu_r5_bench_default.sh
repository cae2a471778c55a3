#!/bin/bash
# the driver's command (`python bench.py`, defaults) with its wall time, and the `secondary` block it now carries
export OVQA_NO_BUILD=1 PYTHONDONTWRITEBYTECODE=1
mkdir -p gpurun_out
t0=$(date +%s)
python bench.py 2> gpurun_out/bench_default.err > gpurun_out/bench_default.json; rc=$?
echo "rc=$rc wall=$(( $(date +%s) - t0 ))s"
tail -3 gpurun_out/bench_default.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/bench_default.json"))
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["cpu_baseline"]["value"])
print(json.dumps(d["secondary"], indent=1)[:3500])
PY
