#!/bin/bash
# round 4, first call: the decode-path tests (G16, replay, masked vocabulary), then a short bench line and the decode lines
set -o pipefail
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
rm -f gpurun_out/parity_r4_decode.tsv
OVQA_PARITY_REPORT=gpurun_out/parity_r4_decode.tsv timeout -k 10 500 python -m pytest tests/test_blocks_gpu.py tests/test_kernels_gpu.py -q -m gpu -p no:cacheprovider -x -k "beam or decode or decoder or stateful or masked" > gpurun_out/tests_decode.log 2>&1 || { echo "tests failed"; grep -E "^(FAILED|ERROR)|Error" gpurun_out/tests_decode.log | head; tail -30 gpurun_out/tests_decode.log; exit 1; }
tail -1 gpurun_out/tests_decode.log
cat gpurun_out/parity_r4_decode.tsv | grep -E "G16|replay" | cut -c1-200
timeout -k 10 300 python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-roofline > gpurun_out/bench_r4_first.json 2> gpurun_out/bench_r4_first.err || { echo bench failed; tail -5 gpurun_out/bench_r4_first.err; exit 1; }
python -c "
import json; r=json.loads(open('gpurun_out/bench_r4_first.json').read().strip().splitlines()[-1]); print('STEP', r['ms_per_step'], r['value'])"
for b in 3 1; do
timeout -k 10 300 python bench.py --workload decode --beam $b --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/bench_r4_decode_b$b.json 2> gpurun_out/bench_r4_decode_b$b.err || { echo decode bench failed; tail -5 gpurun_out/bench_r4_decode_b$b.err; exit 1; }
python -c "
import json; r=json.loads(open('gpurun_out/bench_r4_decode_b$b.json').read().strip().splitlines()[-1]); print('DECODE beam $b', r['ms_per_step'], r['value'], r.get('unit'))"
done
