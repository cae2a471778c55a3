#!/bin/bash
# HBM traffic of the step's kernels from the TCC counters (two separate --pmc passes, as
# MI355X_MICROARCH.md "rocprofv3 PMC slots" prescribes: FETCH_SIZE needs 3 TCC slots, WRITE_SIZE 2).
mkdir -p gpurun_out/traffic; rm -rf gpurun_out/traffic/*
export PYTHONDONTWRITEBYTECODE=1
# never compile from a profiled run (hipcc under the profiler preload would be an exec after GPU init): build first
python -m openvivqa_amd.build > /dev/null 2>&1 || { echo "library build failed"; exit 1; }
export OVQA_NO_BUILD=1
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/traffic/$c -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > $R/gpurun_out/traffic/$c.log 2>&1
  echo "$c exit $?"
done
python3 - <<'PY'
import csv, glob, collections, json, os
root = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/traffic"
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(f"{root}/{c}/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == c:
                acc[r["Kernel_Name"]][c].append(float(r["Counter_Value"]))
out = {}
for k, v in acc.items():
    if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
        f, w = sum(v["FETCH_SIZE"]) / len(v["FETCH_SIZE"]), sum(v["WRITE_SIZE"]) / len(v["WRITE_SIZE"])
        # gfx950: FETCH_SIZE counts 64 B per 128-B request of a wide coalesced read -> x2; both are in KB
        out[k] = {"launches": len(v["FETCH_SIZE"]), "fetch_kb_avg": f, "write_kb_avg": w,
                  "hbm_bytes_per_launch": (2 * f + w) * 1024}
json.dump(out, open(root + "/traffic.json", "w"), indent=1)
for k, v in sorted(out.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * kv[1]["launches"])[:14]:
    print(f'{k[:90]:90s} n={v["launches"]:4d} fetch={v["fetch_kb_avg"]/1024:8.2f}MB(x2) write={v["write_kb_avg"]/1024:8.2f}MB')
PY
