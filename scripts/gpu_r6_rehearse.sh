#!/bin/bash
# Round 6: single-rank RCCL rehearsal of the N > 1 machinery on one box -- plain N = 1 step, the replicated optimiser behind
# the default segmentation, the sharded optimiser with the slice sizes of 8 ranks (timing only: see bench.py --rehearse-shard).
#   -> gpurun_out/r06_rehearse_{plain,seg5,seg5_sharded}.json
python -m openvivqa_amd.build > /dev/null 2>&1 || { echo "library build failed"; exit 1; }
export OVQA_NO_BUILD=1
mkdir -p gpurun_out
B="python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-roofline --no-secondary --repeats 3"
for r in 1 2; do
  $B --no-fuse-adam > gpurun_out/r06_rehearse_plain.json 2> gpurun_out/r06_rehearse_plain.err
  $B --rehearse-comm --no-shard > gpurun_out/r06_rehearse_seg5.json 2> gpurun_out/r06_rehearse_seg5.err
  $B --rehearse-comm --rehearse-shard 8 > gpurun_out/r06_rehearse_seg5_sharded.json 2> gpurun_out/r06_rehearse_seg5_sharded.err
  for f in plain seg5 seg5_sharded; do
    python - gpurun_out/r06_rehearse_$f.json $f <<'PY'
import json, sys
r = json.load(open(sys.argv[1]))
print(sys.argv[2], r["ms_per_step_median"], r["ms_per_step_min"], r["config"]["grad_segments"], r["final_loss"])
PY
  done
done
