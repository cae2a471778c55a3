#!/bin/bash
# beam-3 decode at full size: eager first, graph only if that is clean
set -o pipefail
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
python -m openvivqa_amd.build > /dev/null 2>&1 || { echo "library build failed"; exit 1; }
export OVQA_NO_BUILD=1
timeout -k 10 300 python -m pytest tests/test_blocks_gpu.py -q -p no:cacheprovider -x -k "beam" > gpurun_out/decode_tests.log 2>&1 || { echo "beam tests failed"; tail -5 gpurun_out/decode_tests.log; exit 1; }
tail -1 gpurun_out/decode_tests.log
AMD_LOG_LEVEL=0 timeout -k 10 120 python bench.py --workload decode --beam 3 --steps 5 --warmup 1 --no-graph 2> gpurun_out/decode_b3_eager.err | cut -c1-700 || { echo "EAGER beam 3 failed"; tail -3 gpurun_out/decode_b3_eager.err; exit 1; }
timeout -k 10 120 python bench.py --workload decode --beam 3 --steps 10 --warmup 2 2> gpurun_out/decode_b3.err | cut -c1-900 || { echo "GRAPH beam 3 failed"; tail -3 gpurun_out/decode_b3.err; exit 1; }
timeout -k 10 120 python bench.py --workload decode --beam 1 --steps 10 --warmup 2 2> gpurun_out/decode_b1.err | cut -c1-900
