#!/bin/bash
# round 4: bench contract tests + PPO tests, then the driver's command and the default one
cd /root/repo
mkdir -p gpurun_out/r4_a
timeout -k 10 1100 python -m pytest tests/test_bench_contract_gpu.py tests/test_ppo.py tests/test_policy_gpu.py -x -q -m gpu > gpurun_out/r4_a/bench_tests.log 2>&1; rc=$?
tail -15 gpurun_out/r4_a/bench_tests.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r4_a/bench_steps20.json 2> gpurun_out/r4_a/bench_steps20.err
timeout -k 10 300 python bench.py > gpurun_out/r4_a/bench_default.json 2> gpurun_out/r4_a/bench_default.err
python - <<'PY'
import json
for f in ("bench_steps20", "bench_default"):
    d=json.loads(open('gpurun_out/r4_a/%s.json' % f).read().strip().split('\n')[-1])
    r=d['roofline']
    print('%s: value %.4g, ms_per_step %.5f (median %.5f), events %.2f us, frac %.3f (%s), one launch %.2f us' % (f, d['value'], d['ms_per_step'], d['ms_per_step_median'], r['launch_us_events'], r['frac'], r['bound'], r['one_launch_us']))
    for k,v in r['configs'].items(): print('   ', k, {a:(round(b,4) if isinstance(b,float) else b) for a,b in v.items() if b is not None and a not in ('finite','feasible_frac')})
PY
