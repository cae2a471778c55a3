#!/bin/bash
# round 4: the whole GPU suite (with the parity report) + smoke + a bench line
set -o pipefail
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
rm -f gpurun_out/parity_report.tsv
OVQA_PARITY_REPORT=gpurun_out/parity_report.tsv timeout -k 10 1000 python -m pytest tests -q -m gpu -p no:cacheprovider ${PYTEST_X:--x} > gpurun_out/tests.log 2>&1 || { echo "tests failed"; grep -E "^(FAILED|ERROR)" gpurun_out/tests.log | head -20; tail -40 gpurun_out/tests.log; exit 1; }
tail -1 gpurun_out/tests.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/smoke.log 2>&1 || { echo "smoke failed"; tail -5 gpurun_out/smoke.log; exit 1; }
tail -1 gpurun_out/smoke.log
timeout -k 10 300 python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-roofline > gpurun_out/bench_quick.json 2> gpurun_out/bench_quick.err || { echo bench failed; tail -5 gpurun_out/bench_quick.err; exit 1; }
python -c "
import json; r=json.loads(open('gpurun_out/bench_quick.json').read().strip().splitlines()[-1]); print('STEP', r['ms_per_step'], r.get('ms_per_step_median'), r['value'])"
# the N > 1 control flow on one rank (RCCL process group alive: captured exchange, timed eager step, then the roofline probes'
# own graph captures next to the group's watchdog thread)
timeout -k 10 400 python bench.py --rehearse-comm --steps 50 --warmup 10 --no-cpu-baseline > gpurun_out/bench_rehearse.json 2> gpurun_out/bench_rehearse.err || { echo "rehearse bench failed"; tail -8 gpurun_out/bench_rehearse.err; exit 1; }
python -c "
import json; r=json.loads(open('gpurun_out/bench_rehearse.json').read().strip().splitlines()[-1]); print('REHEARSE', r['ms_per_step'], r['config']['grad_segments'], r['gradient_exchange']['exposed_ms_total'], 'roofline', r['roofline']['frac'])"
