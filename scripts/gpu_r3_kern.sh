#!/bin/bash
mkdir -p gpurun_out
python -m openvivqa_amd.build > /dev/null 2>&1 || exit 1
export OVQA_NO_BUILD=1
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu > gpurun_out/kern_test.log 2>&1; rc=$?; echo "kernel tests exit $rc"; tail -3 gpurun_out/kern_test.log
[ $rc -eq 0 ] || grep -E "^E |^FAILED" gpurun_out/kern_test.log | head -20
