#!/bin/bash
# decode path after the glue kernels + skinny GEMMs: tests (decode + every GEMM consumer), bench, profile
set -o pipefail
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
python -m openvivqa_amd.build > /dev/null 2>&1 || { echo "library build failed"; exit 1; }
export OVQA_NO_BUILD=1
timeout -k 10 900 python -m pytest tests/test_kernels_gpu.py tests/test_blocks_gpu.py -q -m gpu -p no:cacheprovider -x > gpurun_out/decode_tests.log 2>&1 || { echo "tests failed"; grep -E "^(FAILED|ERROR)|Error" gpurun_out/decode_tests.log | head; tail -5 gpurun_out/decode_tests.log; exit 1; }
tail -1 gpurun_out/decode_tests.log
for beam in 3 1; do
  timeout -k 10 120 python bench.py --workload decode --beam $beam --steps 20 --warmup 2 2> gpurun_out/decode_b$beam.err | cut -c1-1100 || { echo "decode bench beam $beam failed"; tail -3 gpurun_out/decode_b$beam.err; exit 1; }
done
OVQA_GEMM_SKINNY_MAXROWS=0 timeout -k 10 120 python bench.py --workload decode --beam 3 --steps 20 --warmup 2 2> gpurun_out/decode_b3_noskinny.err | cut -c1-400
timeout -k 10 200 python bench.py --workload m4c_decode --steps 5 --warmup 2 2> gpurun_out/m4c.err | cut -c1-600
bash scripts/gpu_r3_decode_prof.sh
