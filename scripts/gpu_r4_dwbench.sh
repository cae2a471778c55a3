#!/bin/bash
# The grouped dW launch of the MCAN step on its own: 128 x 128 tiles against 256 x 256 tiles (scripts/dw_bench.py).
set -o pipefail
export OVQA_NO_BUILD=1
for v in 0 1; do OVQA_DW_TILE256=$v timeout -k 10 200 python scripts/dw_bench.py 2>&1 | tail -1; done
