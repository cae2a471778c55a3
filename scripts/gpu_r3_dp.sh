#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
python -m openvivqa_amd.build > /dev/null 2>&1 || { echo "library build failed"; exit 1; }
export OVQA_NO_BUILD=1
rm -f gpurun_out/parity_dp.tsv
OVQA_PARITY_REPORT=$PWD/gpurun_out/parity_dp.tsv timeout -k 10 600 python -m pytest tests/test_train_gpu.py -q -p no:cacheprovider -x -k "data_parallel_exchange" > gpurun_out/dp_test.log 2>&1
echo "dp test exit $?"; tail -5 gpurun_out/dp_test.log; cat gpurun_out/parity_dp.tsv
# the contract form of the launcher on this one-GPU box: refuses N > devices; gloo rehearsal starts N real ranks


