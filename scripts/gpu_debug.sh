#!/bin/bash
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
timeout -k 10 300 python scripts/debug_defer.py > gpurun_out/debug.log 2>&1; echo "exit $?"; grep -v amdgpu.ids gpurun_out/debug.log | tail -40
