#!/bin/bash
# single-rank rehearsal of the N > 1 code path (RCCL group of one: the all-reduce is a copy -> prices the machinery)
mkdir -p gpurun_out
python -m openvivqa_amd.build > /dev/null 2>&1 || { echo "library build failed"; exit 1; }
export OVQA_NO_BUILD=1 PYTHONDONTWRITEBYTECODE=1
run() {
  timeout -k 10 200 python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-roofline --repeats 3 "$@" 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$*', '| ms/step', d['ms_per_step'], d['ms_per_step_median'], '| segments', d['config']['grad_segments'], '|', json.dumps(d.get('gradient_exchange'))[:300])
" || return 1
}
run && run --rehearse-comm --comm-dtype fp32 && run --rehearse-comm --comm-dtype bf16 && run --rehearse-comm --comm-dtype fp32 --overlap-mb 48 && run --rehearse-comm --comm-dtype fp32 --overlap-mb 0
