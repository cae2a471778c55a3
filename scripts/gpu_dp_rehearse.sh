#!/bin/bash
# never compile from a profiled run (hipcc under the profiler preload would be an exec after GPU init): build first
python -m openvivqa_amd.build > /dev/null 2>&1 || { echo "library build failed"; exit 1; }
export OVQA_NO_BUILD=1
# GPU: harness tests, then the bench in its normal N=1 form and with the N>1 code path rehearsed on one rank.
set -e
mkdir -p gpurun_out
python -m pytest tests/test_train_gpu.py tests/test_kernels_gpu.py -x -q -k "train or cast" 2>&1 | tee gpurun_out/train_tests.log | tail -5
python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-roofline > gpurun_out/bench_plain.json 2> gpurun_out/bench_plain.err
cat gpurun_out/bench_plain.json
for ov in 32 0; do
python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-roofline --rehearse-comm --overlap-mb $ov > gpurun_out/bench_rehearse_$ov.json 2> gpurun_out/bench_rehearse_$ov.err
cat gpurun_out/bench_rehearse_$ov.json
done
cd /tmp && export TMPDIR=/tmp
for ov in 32 0; do
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_reh_$ov -o reh -- python3 $GRAFT_REPO_ROOT/bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-roofline --rehearse-comm --overlap-mb $ov > /dev/null 2>&1
done
ls $GRAFT_REPO_ROOT/gpurun_out/prof_reh_32
