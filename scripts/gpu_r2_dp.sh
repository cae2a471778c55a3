#!/bin/bash
# N>1 code path on one GPU: (1) single-rank RCCL rehearsal (phased backward, comm stream, bf16 transport) with the
# per-segment exchange timing, (2) four real ranks on GPU 0 over gloo (cross-process control flow: segment plan agreement,
# barriers, max-over-ranks timing) -- the box allows six processes on its card, launcher included.
set -o pipefail
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
python -m openvivqa_amd.build > /dev/null 2>&1 || { echo "library build failed"; exit 1; }
export OVQA_NO_BUILD=1
B="--steps 100 --warmup 10 --no-cpu-baseline --no-roofline --repeats 3"
timeout -k 10 200 python bench.py $B > gpurun_out/dp_plain.json 2> gpurun_out/dp_plain.err; echo "plain exit $?"
timeout -k 10 200 python bench.py $B --rehearse-comm > gpurun_out/dp_reh96.json 2> gpurun_out/dp_reh96.err; echo "rehearse 96 exit $?"
timeout -k 10 200 python bench.py $B --rehearse-comm --overlap-mb 0 > gpurun_out/dp_reh0.json 2> gpurun_out/dp_reh0.err; echo "rehearse 0 exit $?"
for f in dp_plain dp_reh96 dp_reh0; do
  python3 -c "
import json,sys
d=json.loads(open('gpurun_out/$f.json').read().strip().splitlines()[-1])
print('$f', d['ms_per_step'], d['ms_per_step_median'], d['config']['grad_segments'], json.dumps(d.get('gradient_exchange')))"
done
OVQA_REHEARSE_BACKEND=gloo timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 4 --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --repeats 1 > gpurun_out/dp4_gloo.json 2> gpurun_out/dp4_gloo.err; echo "gloo 4 ranks exit $?"
tail -1 gpurun_out/dp4_gloo.json | cut -c1-900
