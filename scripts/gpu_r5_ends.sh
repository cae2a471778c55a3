#!/bin/bash
# round 5: the model-end kernels (embedding rows, dropout, attention pooling, log_softmax + NLL), then the models that use them
set -o pipefail
export OVQA_NO_BUILD=1 PYTHONDONTWRITEBYTECODE=1
mkdir -p gpurun_out
timeout -k 10 400 python -m pytest tests/test_kernels_gpu.py -x -q -k "embed or dropout_apply or attention_pool or log_softmax or lstm" 2>&1 | tail -25 || exit 1
timeout -k 10 600 python -m pytest tests/test_modules_gpu.py -x -q -k "G17 or G12 or G7 or G16 or manifest" 2>&1 | tail -25 || exit 1
for w in model cross_modality; do
  timeout -k 10 300 python bench.py --workload $w --steps 30 --warmup 5 --no-cpu-baseline --no-roofline 2> gpurun_out/bench_$w.err | cut -c1-300 || { tail -5 gpurun_out/bench_$w.err; exit 1; }
done
