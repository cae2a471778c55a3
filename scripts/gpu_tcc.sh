#!/bin/bash
# L2 behaviour per kernel: TCC hit / miss / requests (separate --pmc pass, kernel trace only)
set -o pipefail
python -m openvivqa_amd.build > /dev/null 2>&1 || { echo "library build failed"; exit 1; }
export OVQA_NO_BUILD=1 PYTHONDONTWRITEBYTECODE=1
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_tcc
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --no-cpu-baseline --no-roofline --repeats 1 --steps 3 --warmup 1"
i=0
for set in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_TAG_STALL_sum TCC_BUSY_sum"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/s$i -- $B > $OUT/s$i.log 2>&1; echo "set $i exit $?"
done
python3 - "$OUT" <<'PY'
import csv, glob, collections, sys, json
out=sys.argv[1]
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out+"/s*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        n=r["Kernel_Name"].replace("(anonymous namespace)::","").replace("void ","")
        acc[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
rows=[]
for k,v in acc.items():
    d={c:sum(x)/len(x) for c,x in v.items()}
    d["n"]=max(len(x) for x in v.values())
    rows.append((k,d))
json.dump({k:d for k,d in rows}, open(out+"/tcc_summary.json","w"), indent=1)
for k,d in sorted(rows,key=lambda kd:-kd[1].get("TCC_REQ_sum",0)*kd[1]["n"])[:14]:
    hit=d.get("TCC_HIT_sum",0); miss=d.get("TCC_MISS_sum",0)
    print(f"{k[:70]:70s} n={d['n']:4d} req={d.get('TCC_REQ_sum',0)/1e6:7.2f}M hit%={100*hit/max(1,hit+miss):5.1f} tcp_lat/req={d.get('TCP_TCC_READ_REQ_LATENCY_sum',0)/max(1,d.get('TCP_TCC_READ_REQ_sum',1)):8.1f} tagstall={d.get('TCC_TAG_STALL_sum',0)/1e6:6.2f}M")
PY
