#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1 OVQA_NO_BUILD=1
timeout -k 10 600 python -m pytest tests/test_train_gpu.py tests/test_kernels_gpu.py -q -m gpu -p no:cacheprovider -x -k "loss or fresh_processes or graph_replay or whole_step" 2>&1 | tail -3
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_loss
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_loss -- python3 $R/bench.py --no-cpu-baseline --no-roofline --repeats 1 --steps 10 --warmup 2 > $R/gpurun_out/prof_loss.log 2>&1
find $R/gpurun_out/prof_loss -name "*kernel_stats.csv" | head -1 | xargs grep -E "sq_loss|begin_step|reduce" | cut -c1-200
cd $R
timeout -k 10 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-roofline --repeats 3 2>/dev/null | python -c "
import sys, json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('STEP', r['ms_per_step'], r.get('ms_per_step_median'), r['value'])"
