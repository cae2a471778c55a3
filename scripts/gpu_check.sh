#!/bin/bash
# First-contact GPU run: kernel tests, module tests, smoke, short bench.  Logs to gpurun_out/.
set -o pipefail
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
echo "== rocminfo ==" > gpurun_out/env.log
(rocminfo | grep -E "Marketing Name|gfx9|Compute Unit" | head -8; nproc; python -c "import torch;print(torch.__version__, torch.cuda.get_device_name(0))") >> gpurun_out/env.log 2>&1
echo "== kernels ==" 
timeout -k 10 420 python -m pytest tests/test_kernels_gpu.py -q -m gpu -p no:cacheprovider > gpurun_out/kernels.log 2>&1
echo "kernel tests exit $?"; tail -5 gpurun_out/kernels.log
echo "== modules =="
timeout -k 10 420 python -m pytest tests/test_modules_gpu.py -q -m gpu -p no:cacheprovider > gpurun_out/modules.log 2>&1
echo "module tests exit $?"; tail -5 gpurun_out/modules.log
echo "== smoke =="
timeout -k 10 200 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1
echo "smoke exit $?"; tail -4 gpurun_out/smoke.log
echo "== bench =="
timeout -k 10 300 python bench.py --steps 10 --warmup 3 > gpurun_out/bench.log 2>&1
echo "bench exit $?"; tail -3 gpurun_out/bench.log
