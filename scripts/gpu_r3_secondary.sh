#!/bin/bash
mkdir -p gpurun_out
python -m openvivqa_amd.build > /dev/null 2>&1 || exit 1
export OVQA_NO_BUILD=1
for w in model cross_modality; do
  timeout -k 10 300 python bench.py --workload $w --steps 20 --warmup 5 2> gpurun_out/bench_$w.err | cut -c1-330 || { echo "$w failed"; tail -3 gpurun_out/bench_$w.err; }
done
timeout -k 10 300 python bench.py --dtype fp32 --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --repeats 1 2> gpurun_out/bench_fp32.err | cut -c1-330
