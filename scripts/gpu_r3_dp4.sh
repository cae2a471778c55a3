#!/bin/bash
# four real ranks on GPU 0 over gloo, started by bench.py's own launcher (the contract form `python bench.py --gpus N`)
set -o pipefail
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
python -m openvivqa_amd.build > /dev/null 2>&1 || { echo "library build failed"; exit 1; }
for cd in fp32 bf16; do
  OVQA_REHEARSE_BACKEND=gloo timeout -k 10 400 python bench.py --gpus 4 --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --repeats 1 --comm-dtype $cd > gpurun_out/dp4_$cd.json 2> gpurun_out/dp4_$cd.err; echo "gloo 4 ranks $cd exit $?"
  tail -1 gpurun_out/dp4_$cd.json | cut -c1-1100
done
