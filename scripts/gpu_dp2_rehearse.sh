#!/bin/bash
# N real ranks (default 2, at most 4) on ONE GPU with gloo transport: exercises the N>1 control flow of bench.py /
# TrainStep across processes -- phased graphs, comm stream, segment order, barrier + max-over-ranks timing.
# Not a performance run.
set -e
N=${1:-2}
mkdir -p gpurun_out
export OVQA_REHEARSE_BACKEND=gloo
timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 29533 \
  bench.py --gpus $N --steps 10 --warmup 3 --no-roofline > gpurun_out/dp$N.json 2> gpurun_out/dp$N.err
tail -1 gpurun_out/dp$N.json | cut -c1-900
