#!/bin/bash
# Round 3: the whole GPU suite, then the default bench line
set -o pipefail
cd /root/repo
OUT=/root/repo/gpurun_out
mkdir -p $OUT
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -8 $OUT/pytest_gpu.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 600 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.load(open('/root/repo/gpurun_out/bench_default.json'))
print('HEAD', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['bound'], d['roofline']['launch_us_events'])
for a in d['also']:
    r=a.get('roofline') or {}
    print(a['workload'], a.get('kernel'), 'value %.3e'%a.get('value',0), 'us', round(a.get('launch_us_events',0) or 0,2), 'frac', round(r.get('frac',0) or 0,4), r.get('bound'), 'hbm', round((r.get('hbm') or {}).get('frac',0),4), 'valu', round(((r.get('valu') or {}) or {}).get('frac',0) or 0,4), a.get('error',''))
print('cpu', d['cpu_baseline']['value'], d['cpu_baseline']['cores'], (d['cpu_baseline'].get('python_env_processes') or {}).get('value'))
PY
