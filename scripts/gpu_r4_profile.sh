#!/bin/bash
# round 4: profiles for profiles/ (kernel stats, HBM traffic, SQ counters of the step) + the default bench line
bash scripts/gpu_profile_step.sh r04 2>&1 | tail -30
cd $GRAFT_REPO_ROOT
export PYTHONDONTWRITEBYTECODE=1 OVQA_NO_BUILD=1
timeout -k 10 600 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err || { echo "bench failed"; tail -8 gpurun_out/bench_default.err; exit 1; }
python - <<'PY'
import json
r = json.loads(open('gpurun_out/bench_default.json').read().strip().splitlines()[-1])
print('STEP', r['ms_per_step'], r.get('ms_per_step_median'), r['value'], 'x cpu', r.get('speedup_vs_cpu_baseline'))
print('roofline', {k: r['roofline'][k] for k in ('achieved', 'frac', 'traffic', 'launches_per_step', 'avg_launch_us')})
for k, v in r['roofline_attention']['kernels'].items():
    print(k, {a: v[a] for a in ('launches_per_step', 'avg_launch_us', 'frac_of_hbm_peak', 'frac_of_mfma_peak', 'mfma_busy_frac')})
print('cpu', {k: r['cpu_baseline'][k] for k in ('value', 'cores', 'kind')})
PY
