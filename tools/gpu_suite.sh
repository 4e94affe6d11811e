#!/bin/bash
# The whole GPU suite, then the driver's exact bench command (`python3 bench.py --gpus 1 --steps 20 --warmup 5`).
#   gpurun --timeout 1200 -- ./tools/gpu_suite.sh [tag]     -> gpurun_out/<tag>/{gpu_suite.log,bench_steps20.json,.err,bench_also.json}
# SKIP_SUITE=1 runs the bench command alone; PYTEST_ARGS narrows the suite (e.g. "-k range").
cd /root/repo
TAG=${1:-r5_a}
OUT=gpurun_out/$TAG
mkdir -p $OUT
if [ -z "$SKIP_SUITE" ]; then
  timeout -k 10 1000 python -m pytest tests -x -q -m gpu $PYTEST_ARGS > $OUT/gpu_suite.log 2>&1; rc=$?
  tail -25 $OUT/gpu_suite.log
  [ $rc -ne 0 ] && exit $rc
fi
[ -n "$SKIP_BENCH" ] && exit 0
timeout -k 10 400 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_steps20.json 2> $OUT/bench_steps20.err || { tail -20 $OUT/bench_steps20.err; exit 1; }
cp bench_also.json $OUT/bench_also.json
python3 - $OUT <<'PY'
import json, sys
out = sys.argv[1]
text = open(out + '/bench_steps20.json').read()
lines = [l for l in text.splitlines() if l.strip()]
print('stdout: %d line(s), %d bytes' % (len(lines), len(lines[-1])))
d = json.loads(lines[-1])
r = d['roofline']
print('steps 20: value %.4g, ms_per_step %.5f (median %.5f), events %.2f us, frac %.3f (%s), one launch %.2f us'
      % (d['value'], d['ms_per_step'], d['ms_per_step_median'], r['launch_us_events'], r['frac'], r['bound'], r['one_launch_us'] or 0))
for w, row in r['configs'].items():
    print('   %-34s %s' % (w, row))
print('   traffic/algorithmic', r['traffic_over_algorithmic'], r.get('traffic_over_algorithmic_rows'), 'frac_wall', r.get('frac_wall'))
print('   cpu', d['cpu_baseline'])
PY
