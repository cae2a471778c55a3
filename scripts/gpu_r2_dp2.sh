#!/bin/bash
# train-step GPU tests, then the single-rank rehearsal of the N > 1 path (segment-wise Adam under the exchange)
set -o pipefail
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
python -m openvivqa_amd.build > /dev/null 2>&1 || { echo "library build failed"; exit 1; }
export OVQA_NO_BUILD=1
timeout -k 10 600 python -m pytest tests/test_train_gpu.py -q -m gpu -p no:cacheprovider -x > gpurun_out/train_test.log 2>&1
rc=$?; echo "train tests exit $rc"; tail -3 gpurun_out/train_test.log
[ $rc -eq 0 ] || { grep -E "^(FAILED|ERROR|E )" gpurun_out/train_test.log | head -30; exit 1; }
B="--steps 100 --warmup 10 --no-cpu-baseline --no-roofline --repeats 3"
timeout -k 10 200 python bench.py $B > gpurun_out/dp_plain.json 2> gpurun_out/dp_plain.err; echo "plain exit $?"
timeout -k 10 200 python bench.py $B --rehearse-comm > gpurun_out/dp_reh96.json 2> gpurun_out/dp_reh96.err; echo "rehearse 96 exit $?"
timeout -k 10 200 python bench.py $B --rehearse-comm --overlap-mb 0 > gpurun_out/dp_reh0.json 2> gpurun_out/dp_reh0.err; echo "rehearse 0 exit $?"
for f in dp_plain dp_reh96 dp_reh0; do
  python3 -c "
import json,sys
d=json.loads(open('gpurun_out/$f.json').read().strip().splitlines()[-1])
print('$f', d['ms_per_step'], d['ms_per_step_median'], d['config']['grad_segments'], json.dumps(d.get('gradient_exchange')))"
done
