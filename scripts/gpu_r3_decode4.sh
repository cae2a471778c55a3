#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
python -m openvivqa_amd.build > /dev/null 2>&1 || { echo "library build failed"; exit 1; }
export OVQA_NO_BUILD=1
timeout -k 10 600 python -m pytest tests -q -m gpu -p no:cacheprovider -x -k "decode or decoder or stateful or beam or reorder or topk" > gpurun_out/decode_tests.log 2>&1 || { echo "decode tests failed"; grep -E "^(FAILED|ERROR)|Error" gpurun_out/decode_tests.log | head; tail -5 gpurun_out/decode_tests.log; exit 1; }
tail -1 gpurun_out/decode_tests.log
for beam in 3 1; do
  timeout -k 10 120 python bench.py --workload decode --beam $beam --steps 20 --warmup 2 2> gpurun_out/decode_b$beam.err | cut -c1-1100 || { echo "decode bench beam $beam failed"; tail -3 gpurun_out/decode_b$beam.err; exit 1; }
done
bash scripts/gpu_r3_decode_prof.sh
