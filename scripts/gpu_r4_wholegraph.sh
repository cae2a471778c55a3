#!/bin/bash
# round 4: the whole step as ONE graph (captured exchange + Adam + device schedule): tests, A/B in the step, rehearsal
set -o pipefail
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
timeout -k 10 600 python -m pytest tests/test_train_gpu.py tests/test_modules_gpu.py -q -m gpu -p no:cacheprovider -x -k "train or trajectory or schedule or whole" > gpurun_out/tests_whole.log 2>&1 || { echo "tests failed"; grep -E "^(FAILED|ERROR)" gpurun_out/tests_whole.log | head; tail -40 gpurun_out/tests_whole.log; exit 1; }
tail -1 gpurun_out/tests_whole.log
line() { python -c "
import json,sys; r=json.loads(open('$1').read().strip().splitlines()[-1]); print('$2', r['ms_per_step'], r.get('ms_per_step_median'), r['value'], r.get('gradient_exchange', ''))"; }
for rep in 1 2; do
  for w in 1 0; do
    OVQA_WHOLE_STEP_GRAPH=$w timeout -k 10 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-roofline --repeats 3 > gpurun_out/b_whole$w.json 2> gpurun_out/b_whole$w.err || { echo bench failed; tail -5 gpurun_out/b_whole$w.err; exit 1; }
    line gpurun_out/b_whole$w.json "plain whole=$w"
  done
done
for w in 1 0; do
  for mb in 48 0; do
    OVQA_WHOLE_STEP_GRAPH=$w timeout -k 10 300 python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-roofline --repeats 3 --rehearse-comm --overlap-mb $mb > gpurun_out/b_reh_$w_$mb.json 2> gpurun_out/b_reh_$w_$mb.err || { echo rehearse failed; tail -8 gpurun_out/b_reh_$w_$mb.err; exit 1; }
    line gpurun_out/b_reh_$w_$mb.json "rehearse whole=$w overlap_mb=$mb"
  done
done
