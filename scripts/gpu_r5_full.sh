#!/bin/bash
# round 5: the whole -m gpu suite with the parity report (every compared quantity next to its bar)
export OVQA_NO_BUILD=1 PYTHONDONTWRITEBYTECODE=1
mkdir -p gpurun_out
rm -f gpurun_out/r05_parity_report.tsv
OVQA_PARITY_REPORT=$PWD/gpurun_out/r05_parity_report.tsv timeout -k 10 1150 python -m pytest tests -m gpu -x -q 2>&1 | tail -15
wc -l gpurun_out/r05_parity_report.tsv
