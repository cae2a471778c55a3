#!/bin/bash
# round-3 profiles of the training step + the secondary cross-modality line
bash scripts/gpu_profile_step.sh r03 > gpurun_out/prof_r03.log 2>&1; echo "profile exit $?"
tail -30 gpurun_out/prof_r03.log | cut -c1-200
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python bench.py --workload cross_modality --steps 20 --warmup 5 > gpurun_out/bench_xmod.json 2> gpurun_out/bench_xmod.err; echo "xmod exit $?"
tail -3 gpurun_out/bench_xmod.err; cut -c1-700 gpurun_out/bench_xmod.json
