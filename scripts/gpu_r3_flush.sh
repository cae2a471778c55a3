#!/bin/bash
# grouped dW of the guided stack on a side stream under the question stack's (latency-bound) backward?
mkdir -p gpurun_out
python -m openvivqa_amd.build > /dev/null 2>&1 || exit 1
export OVQA_NO_BUILD=1
for f in off 1536 off 1344 1800; do
  if [ $f = off ]; then unset OVQA_WGRAD_FLUSH_TILES; else export OVQA_WGRAD_FLUSH_TILES=$f; fi
  timeout -k 10 200 python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-roofline --repeats 3 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('flush=$f ms/step', d['ms_per_step'], d['ms_per_step_median'])"
done
