#!/bin/bash
# full GPU suite (with the parity report) + smoke + default bench
set -o pipefail
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
python -m openvivqa_amd.build > /dev/null 2>&1 || { echo "library build failed"; exit 1; }
export OVQA_NO_BUILD=1
rm -f gpurun_out/parity_report.tsv
OVQA_PARITY_REPORT=gpurun_out/parity_report.tsv timeout -k 10 1000 python -m pytest tests -q -m gpu -p no:cacheprovider -x > gpurun_out/tests.log 2>&1 || { echo "tests failed"; grep -E "^(FAILED|ERROR)|Error" gpurun_out/tests.log | head; tail -8 gpurun_out/tests.log; exit 1; }
tail -1 gpurun_out/tests.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/smoke.log 2>&1 || { echo "smoke failed"; tail -5 gpurun_out/smoke.log; exit 1; }
tail -1 gpurun_out/smoke.log
timeout -k 10 600 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err || { echo "bench failed"; tail -5 gpurun_out/bench_default.err; exit 1; }
cut -c1-900 gpurun_out/bench_default.json
