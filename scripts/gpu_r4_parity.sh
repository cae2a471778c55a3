#!/bin/bash
# round 4: the parity items (G16 on the GPU, the non-degenerate operating point of the L=6 stacks, escape-clause census)
set -o pipefail
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
rm -f gpurun_out/parity_r4.tsv
OVQA_PARITY_REPORT=gpurun_out/parity_r4.tsv timeout -k 10 900 python -m pytest tests/test_modules_gpu.py tests/test_blocks_gpu.py -q -m gpu -p no:cacheprovider -k "${PYTEST_K:-stack_forward or config3 or fullsize or beam_search}" > gpurun_out/tests_parity.log 2>&1
rc=$?
tail -25 gpurun_out/tests_parity.log
echo "--- escape-clause rows:"; grep -c "escape clause\]" gpurun_out/parity_r4.tsv
grep -E "escape clause|north-star|passed through|G16" gpurun_out/parity_r4.tsv | cut -c1-220 | tail -60
echo "--- sharp: worst rows"; grep "scores x" gpurun_out/parity_r4.tsv | grep "vs emulation" | sort -t$'\t' -k3 -g -r | head -12 | cut -c1-200
exit $rc
