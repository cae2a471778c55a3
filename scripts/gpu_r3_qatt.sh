#!/bin/bash
# query projection inside the guided / cross attention forward kernel: kernel tests (fused shapes and fall-backs)
mkdir -p gpurun_out
python -m openvivqa_amd.build > /dev/null 2>&1 || exit 1
export OVQA_NO_BUILD=1
timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "attention_q_fwd or delta" > gpurun_out/qatt_tests.log 2>&1; rc=$?; echo "kernel tests exit $rc"; tail -3 gpurun_out/qatt_tests.log
OVQA_FORCE_SIMPLE=1 timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "attention_q_fwd or delta" 2>&1 | tail -1
