#!/bin/bash
# LayerNorm kernels after a change: suites, step time, per-kernel time (rocprofv3)
set -o pipefail
export OVQA_NO_BUILD=1
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py tests/test_blocks_gpu.py -q -x -p no:cacheprovider -k "layernorm or ln or block or ffn or encoder or prologue or pos" 2>&1 | tail -2
for rep in 1 2; do
timeout -k 10 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-roofline --repeats 3 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('STEP', d['ms_per_step'])"
done
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_ln
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_ln -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-roofline --repeats 1 > $R/gpurun_out/lnprof.log 2>&1
f=$(find /tmp/prof_ln -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && { cp $f $R/gpurun_out/ln_kernel_stats.csv; grep -i "ln_" $f | cut -c1-70,150-260; }
