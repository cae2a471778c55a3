#!/bin/bash
# round 4: the GPU suites under the alternate code paths (A/B switches of this and earlier rounds); PART=a|b|c|d splits the
# list over four calls (a call may run 1200 s: six variants of ~3 min)
mkdir -p gpurun_out
export OVQA_NO_BUILD=1 PYTHONDONTWRITEBYTECODE=1
if [ "${PART:-a}" = d ]; then
  LIST=("OVQA_DOBWD_PAIR=1" "OVQA_GEMM_CSTEP=0")
elif [ "${PART:-a}" = c ]; then
  LIST=("OVQA_DW_TILE256=2" "OVQA_DW_TILE256=1")
elif [ "${PART:-a}" = a ]; then
  LIST=("OVQA_FORCE_SIMPLE=1" "OVQA_NO_FUSED_QKV=1" "OVQA_NO_FUSED_Q=1" "OVQA_NO_FUSED_DO=1" "OVQA_QATT_PAIR=0" "OVQA_DEFER_WGRAD=0")
else
  LIST=("OVQA_GEMM_BIG16=0" "OVQA_GEMM_SKINNY_MAXROWS=0" "OVQA_GEMM_KSPLIT=3 OVQA_GEMM_KSPLIT_MINK=512 OVQA_DW_KSPLIT=1" "OVQA_DECODE_SPLIT_MIN=1000" "OVQA_WHOLE_STEP_GRAPH=0" "OVQA_ADAM_TILED=0")
fi
for e in "${LIST[@]}"; do
  echo "== $e"
  env $e timeout -k 10 500 python -m pytest tests/test_kernels_gpu.py tests/test_blocks_gpu.py tests/test_modules_gpu.py tests/test_train_gpu.py -q -x -m gpu -p no:cacheprovider --deselect tests/test_train_gpu.py::test_whole_step_graph_equals_phase_graphs_with_eager_adam 2>&1 | tail -1
done
