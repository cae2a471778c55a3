#!/bin/bash
# does the captured step's launch floor come from the fork/join (side-stream dW) or the memcpy nodes in the graph?
set -o pipefail
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
B="python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-roofline"
for cfg in "0 0" "1 0" "0 1" "1 1" "0 0"; do
  set -- $cfg
  OVQA_WGRAD_SIDE=$1 OVQA_GRAPH_MEMCPY=$2 timeout -k 10 200 $B > gpurun_out/floor_$1_$2.log 2>&1
  echo "side=$1 memcpy=$2 exit $? $(tail -1 gpurun_out/floor_$1_$2.log | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["value"], d["final_loss"])')"
done
timeout -k 10 400 python -m pytest tests/test_train_gpu.py -q -m gpu -p no:cacheprovider -x > gpurun_out/train_tests.log 2>&1; echo "train tests exit $?"; tail -3 gpurun_out/train_tests.log
