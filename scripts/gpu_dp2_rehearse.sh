#!/bin/bash
# Two ranks on ONE GPU (gloo transport): exercises the N>1 control flow of bench.py / TrainStep across real
# processes -- phased graphs, comm stream, segment order, barrier + max-over-ranks timing.  Not a performance run.
set -e
mkdir -p gpurun_out
export OVQA_REHEARSE_BACKEND=gloo
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 \
  bench.py --gpus 2 --steps 20 --warmup 5 --no-roofline > gpurun_out/dp2.json 2> gpurun_out/dp2.err
tail -1 gpurun_out/dp2.json | cut -c1-900
unset OVQA_REHEARSE_BACKEND
python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-roofline --rehearse-comm 2>/dev/null | cut -c1-120
