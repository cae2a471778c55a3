#!/bin/bash
mkdir -p gpurun_out/pmc
export PYTHONDONTWRITEBYTECODE=1
# never compile from a profiled run (hipcc under the profiler preload would be an exec after GPU init): build first
python -m openvivqa_amd.build > /dev/null 2>&1 || { echo "library build failed"; exit 1; }
export OVQA_NO_BUILD=1
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for shape in "6400 512 2048 fwd" "6400 2048 512 fwd" "6400 2048 512 dx"; do
  tag=$(echo $shape | tr ' ' '_')
  for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM"; do
    st=$(echo $set | cut -c1-12 | tr ' ' '_')
    timeout -k 10 120 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $R/gpurun_out/pmc/${tag}_$st -- python3 $R/scripts/one_gemm.py $shape > /dev/null 2>&1
    echo "$tag [$set] exit $?"
  done
done
python3 - <<'PY'
import csv, glob, collections, os
root = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/pmc"
for d in sorted(glob.glob(root + "/*")):
    for f in glob.glob(d + "/*/*counter_collection.csv"):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "gemm" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        print(os.path.basename(d), {k: round(sum(v) / len(v)) for k, v in acc.items()})
PY
