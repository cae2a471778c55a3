#!/bin/bash
mkdir -p gpurun_out/pmc
rm -rf gpurun_out/pmc/*
export PYTHONDONTWRITEBYTECODE=1
# never compile from a profiled run (hipcc under the profiler preload would be an exec after GPU init): build first
python -m openvivqa_amd.build > /dev/null 2>&1 || { echo "library build failed"; exit 1; }
export OVQA_NO_BUILD=1
R=$GRAFT_REPO_ROOT
for s in "100 100" "100 20" "20 20"; do python3 $R/scripts/one_attn.py $s time 2>&1 | grep -v amdgpu; done
cd /tmp && export TMPDIR=/tmp
for shape in "100 100 fwd" "100 100 bwd"; do
  tag=$(echo $shape | tr ' ' '_')
  for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM"; do
    st=$(echo $set | cut -c1-12 | tr ' ' '_')
    timeout -k 10 120 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $R/gpurun_out/pmc/${tag}_$st -- python3 $R/scripts/one_attn.py $shape > /dev/null 2>&1
  done
done
python3 - <<'PY'
import csv, glob, collections, os
root = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/pmc"
for d in sorted(glob.glob(root + "/*")):
    for f in glob.glob(d + "/*/*counter_collection.csv"):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            if "attn" in r["Kernel_Name"]:
                acc[r["Kernel_Name"][:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for kname, c in acc.items():
            print(os.path.basename(d), kname, {k: round(sum(v) / len(v)) for k, v in c.items()})
PY
