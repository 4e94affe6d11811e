#!/bin/bash
# round 4: the whole GPU suite, then the driver's bench command
cd /root/repo
mkdir -p gpurun_out/r4_a
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r4_a/gpu_suite.log 2>&1; rc=$?
tail -25 gpurun_out/r4_a/gpu_suite.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r4_a/bench_steps20.json 2> gpurun_out/r4_a/bench_steps20.err && python - <<'PY'
import json
d=json.loads(open('gpurun_out/r4_a/bench_steps20.json').read().strip().split('\n')[-1])
print('steps 20: value %.4g, ms_per_step %.5f, events %.2f us, frac %.3f (%s)' % (d['value'], d['ms_per_step'], d['roofline']['launch_us_events'], d['roofline']['frac'], d['roofline']['bound']))
for a in d['also']: print('  ', a['workload'], '%.4g' % a['value'], a.get('launch_us_events'), a.get('kernel'))
PY
