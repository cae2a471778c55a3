#!/bin/bash
# attention kernels after a staging change: suites, step time, per-kernel times (rocprofv3)
set -o pipefail
export OVQA_NO_BUILD=1
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_kernels_gpu.py tests/test_blocks_gpu.py tests/test_modules_gpu.py -q -x -p no:cacheprovider -k "attention or attn or mha or block or layer or stack or guided" > gpurun_out/attn_stage_tests.log 2>&1 || { tail -30 gpurun_out/attn_stage_tests.log; exit 1; }
tail -1 gpurun_out/attn_stage_tests.log
for rep in 1 2; do
timeout -k 10 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-roofline --repeats 3 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('STEP', d['ms_per_step'])"
done
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_at
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_at -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-roofline --repeats 1 > $R/gpurun_out/atprof.log 2>&1
f=$(find /tmp/prof_at -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && { cp $f $R/gpurun_out/at_kernel_stats.csv; grep -i "attn_" $f | cut -c1-75,100-190; }
