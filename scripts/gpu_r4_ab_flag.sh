#!/bin/bash
# A/B of a compile-time switch on ONE box: FLAG="-DNAME=0" (the alternative) against the default build, alternated twice.
# usage: FLAG="-DOVQA_EPI_PREFETCH=0" bash scripts/gpu_r4_ab_flag.sh
set -o pipefail
mkdir -p gpurun_out
run() {
  OVQA_NO_BUILD=1 timeout -k 10 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-roofline --repeats 3 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1', d['ms_per_step'])"
}
for rep in 1 2; do
  OVQA_EXTRA_HIPCC_FLAGS="$FLAG" python -m openvivqa_amd.build > gpurun_out/build_alt.log 2>&1 || { tail gpurun_out/build_alt.log; exit 1; }
  OVQA_EXTRA_HIPCC_FLAGS="$FLAG" run "alt($FLAG)"
  python -m openvivqa_amd.build > gpurun_out/build_def.log 2>&1 || exit 1
  run default
done
