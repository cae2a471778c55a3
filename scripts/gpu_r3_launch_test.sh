#!/bin/bash
mkdir -p gpurun_out
python -m openvivqa_amd.build > /dev/null 2>&1 || exit 1
timeout -k 10 700 python -m pytest tests/test_train_gpu.py -x -q -m gpu -k "launcher_two_ranks" > gpurun_out/launch_test.log 2>&1; echo "exit $?"; tail -5 gpurun_out/launch_test.log
