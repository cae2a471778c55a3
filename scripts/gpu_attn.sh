#!/bin/bash
# attention kernels: tests (production build), per-kernel times of the step, the step, then the phase probe build
set -o pipefail
mkdir -p gpurun_out
python -m openvivqa_amd.build > /dev/null 2>&1 || { echo "library build failed"; exit 1; }
export PYTHONDONTWRITEBYTECODE=1
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -q -m gpu -p no:cacheprovider -x -k "attention" > gpurun_out/attn_test.log 2>&1
rc=$?; echo "attention tests exit $rc"; tail -3 gpurun_out/attn_test.log
[ $rc -eq 0 ] || { grep -E "^(FAILED|ERROR|E )" gpurun_out/attn_test.log | head -30; exit 1; }
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_attn
rm -rf $OUT; mkdir -p $OUT
( cd /tmp && export TMPDIR=/tmp OVQA_NO_BUILD=1 && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --no-cpu-baseline --no-roofline --repeats 1 --steps 10 --warmup 2 > $OUT/trace.log 2>&1; echo "trace exit $?" )
find $OUT/trace -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
rm -rf $OUT/trace
python - <<'PY'
import csv
for r in csv.DictReader(open('gpurun_out/prof_attn/kernel_stats.csv')):
    n=r['Name'].replace('(anonymous namespace)::','').replace('void ','')
    if 'attn' in n: print(f"{n[:70]:70s} {int(r['Calls'])/15:5.1f} {float(r['AverageNs'])/1e3:7.2f} min {float(r['MinNs'])/1e3:.2f}")
PY
OVQA_NO_BUILD=1 timeout -k 10 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-roofline --repeats 3 > gpurun_out/bench_now.log 2>&1; echo "bench exit $?"; tail -1 gpurun_out/bench_now.log | cut -c1-260
timeout -k 10 400 python scripts/phase_probe.py 2>/dev/null | grep "image"
