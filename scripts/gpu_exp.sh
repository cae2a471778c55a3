#!/bin/bash
python -m pytest tests/test_kernels_gpu.py -x -q -k "attention" 2>&1 | tail -1
for s in "100 100" "100 20" "20 20"; do python3 scripts/one_attn.py $s time 2>&1 | grep -v amdgpu; done
