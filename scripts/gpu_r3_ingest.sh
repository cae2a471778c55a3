#!/bin/bash
mkdir -p gpurun_out
python -m openvivqa_amd.build > /dev/null 2>&1 || exit 1
timeout -k 10 400 python -m pytest tests/test_modules_gpu.py -x -q -m gpu -k "ingestion or feature_embedding" > gpurun_out/ingest_tests.log 2>&1; echo "tests exit $?"; tail -5 gpurun_out/ingest_tests.log
