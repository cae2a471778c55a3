#!/bin/bash
# Four real ranks of bench.py on ONE GPU over gloo: cross-process control flow of the N>1 path (plan agreement across
# ranks, phased graphs, barriers, max-over-ranks timing, gradient_exchange stats).  Not a performance run.
set -o pipefail
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
python -m openvivqa_amd.build > /dev/null 2>&1 || { echo "library build failed"; exit 1; }
export OVQA_NO_BUILD=1 OVQA_REHEARSE_BACKEND=gloo
timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 4 --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --repeats 2 > gpurun_out/dp4_gloo.json 2> gpurun_out/dp4_gloo.err; echo "gloo 4 ranks exit $?"
tail -1 gpurun_out/dp4_gloo.json | cut -c1-1500; tail -3 gpurun_out/dp4_gloo.err | cut -c1-300
