#!/bin/bash
# The supported alternative paths behind the README's environment switches, each over the part of the GPU suite that reaches it
# (the driver's suite runs the defaults).  One pytest process after the other; logs under gpurun_out/alt_*.log.
set -o pipefail
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
run() {  # run <tag> "<VAR=.. VAR=..>" <pytest args...>
  local tag=$1 envs=$2; shift 2
  ( for e in $envs; do export $e; done
    timeout -k 10 900 python -m pytest -q -m gpu -p no:cacheprovider "$@" > gpurun_out/alt_$tag.log 2>&1 )
  echo "$tag ($envs): exit $? -- $(tail -1 gpurun_out/alt_$tag.log)"
}
run simple      "OVQA_FORCE_SIMPLE=1"        tests/test_kernels_gpu.py tests/test_blocks_gpu.py
run nofused_qkv "OVQA_NO_FUSED_QKV=1"        tests/test_blocks_gpu.py tests/test_modules_gpu.py -k "not m4c and not M4C"
run nofused_q   "OVQA_NO_FUSED_Q=1"          tests/test_blocks_gpu.py -k "guided or Guided or cross or mha or MHA or block"
run nofused_do  "OVQA_NO_FUSED_DO=1"         tests/test_blocks_gpu.py tests/test_train_gpu.py -k "not launcher and not ranks"
# (tests that ASSERT the default path was taken -- Adam inside the grouped launch, the whole-step graph -- are left out below)
run nodefer     "OVQA_DEFER_WGRAD=0"         tests/test_blocks_gpu.py tests/test_train_gpu.py -k "not launcher and not ranks and not adam_inside"
run phasegraph  "OVQA_WHOLE_STEP_GRAPH=0"    tests/test_train_gpu.py -k "not launcher and not ranks and not adam_inside and not whole_step"
run t256off     "OVQA_GEMM_T256=0"           tests/test_kernels_gpu.py -k "linear"
run noprefix    "OVQA_NO_PREFIX_LM=1"        tests/test_blocks_gpu.py tests/test_modules_gpu.py -k "m4c or M4C or mmt or MMT or prefix"
