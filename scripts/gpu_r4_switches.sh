#!/bin/bash
# the run-time A/B switches once more on the round's final kernels (one box, each against the default before and after)
set -o pipefail
export OVQA_NO_BUILD=1
run() {
  env $1 timeout -k 10 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-roofline --repeats 3 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1', d['ms_per_step'])"
}
LIST=${LIST:-"X=default OVQA_QATT_PAIR=0 OVQA_QATT_NBUF=2 OVQA_GEMM_BIG16=4 OVQA_GEMM_BIG16=2 OVQA_ATTN_BWD_MERGED=2 OVQA_QKV_FORM=2 X=default"}
for e in $LIST; do run $e; done
