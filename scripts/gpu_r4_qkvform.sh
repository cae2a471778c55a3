#!/bin/bash
# the fused QKV-projection + attention forward with TWO samples (256 rows) per workgroup
# (OVQA_QKV_FORM=2: flop per fetched byte 78 -> 112) against the default one sample x two workgroups per CU:
# step time (alternated) and the matrix-pipe busy fraction of the kernel from a separate PMC pass.
mkdir -p gpurun_out
export OVQA_NO_BUILD=1 PYTHONDONTWRITEBYTECODE=1
R=$GRAFT_REPO_ROOT
for rep in 1 2; do for f in 1 2; do
OVQA_QKV_FORM=$f timeout -k 10 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-roofline --repeats 3 2>/dev/null | python -c "
import sys, json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('QKV_FORM $f', r['ms_per_step'], r.get('ms_per_step_median'))"
done; done
cd /tmp && export TMPDIR=/tmp
for f in 1 2; do
rm -rf $R/gpurun_out/qkvform$f
OVQA_QKV_FORM=$f timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY --output-format csv -d $R/gpurun_out/qkvform$f -- python3 $R/bench.py --no-cpu-baseline --no-roofline --repeats 1 --steps 3 --warmup 1 > $R/gpurun_out/qkvform$f.log 2>&1
python3 - $R/gpurun_out/qkvform$f $f <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "attn_qkv_fwd_mfma_kernel<128" in r["Kernel_Name"]:
            acc[r["Kernel_Name"].split("(")[0][-60:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    m = {c: sum(x) / len(x) for c, x in v.items()}
    print("form", sys.argv[2], k, "launches", len(v["GRBM_GUI_ACTIVE"]), "mfma_busy_frac", round(m["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * m["GRBM_GUI_ACTIVE"] / 8.0), 4),
          "wait_frac", round(m["SQ_WAIT_ANY"] / m["SQ_WAVE_CYCLES"], 3))
PY
done
