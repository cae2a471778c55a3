#!/bin/bash
set -o pipefail
export OVQA_NO_BUILD=1 PYTHONDONTWRITEBYTECODE=1
timeout -k 10 900 python -m pytest tests/test_train_gpu.py -x -q 2>&1 | tail -15
