#!/bin/bash
# round 4: the secondary workloads once (whole models through the one-graph step, fp32 mode, decode, M4C decode)
mkdir -p gpurun_out
export OVQA_NO_BUILD=1 PYTHONDONTWRITEBYTECODE=1
for w in model cross_modality; do
  timeout -k 10 300 python bench.py --workload $w --steps 50 --warmup 5 --no-cpu-baseline --no-roofline 2> gpurun_out/bench_$w.err | cut -c1-400 || { echo "$w failed"; tail -3 gpurun_out/bench_$w.err; }
done
timeout -k 10 300 python bench.py --dtype fp32 --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --repeats 1 2> gpurun_out/bench_fp32.err | cut -c1-330
for beam in 3 1; do
  timeout -k 10 120 python bench.py --workload decode --beam $beam --steps 30 --warmup 3 --no-cpu-baseline 2> gpurun_out/decode_b$beam.err | cut -c1-900
done
timeout -k 10 200 python bench.py --workload m4c_decode --steps 5 --warmup 2 --no-cpu-baseline 2> gpurun_out/m4c.err | cut -c1-500
