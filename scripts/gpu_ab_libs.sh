#!/bin/bash
# A/B of two builds of the library on ONE box: scripts/_build/lib_old.so and lib_new.so are copied over
# openvivqa_amd/csrc/libovqa_hip.so in turn (alternated), the training step timed after each copy.
#   usage: bash scripts/gpu_ab_libs.sh [rounds] [extra bench.py args]
R=${1:-2}; shift
for r in $(seq 1 $R); do
  for v in old new; do
    cp scripts/_build/lib_$v.so openvivqa_amd/csrc/libovqa_hip.so
    timeout -k 10 300 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-roofline --no-secondary --repeats 3 "$@" 2>/dev/null | python -c "
import sys, json
r=json.loads(sys.stdin.read()); print('$v', r['ms_per_step_median'], r['ms_per_step_min'], r['final_loss'])"
  done
done
cp scripts/_build/lib_new.so openvivqa_amd/csrc/libovqa_hip.so
