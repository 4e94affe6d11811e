#!/bin/bash
# the default bench command (400-step regions) and a few single-workload lines named on the command line
#   gpurun -- ./tools/gpu_bench_default.sh <tag> ["<bench.py args>" ...]
cd /root/repo
TAG=${1:-r5_a}; shift
mkdir -p gpurun_out/$TAG
timeout -k 10 500 python3 bench.py > gpurun_out/$TAG/bench_default.json 2> gpurun_out/$TAG/bench_default.err || { tail -5 gpurun_out/$TAG/bench_default.err; exit 1; }
cp bench_also.json gpurun_out/$TAG/bench_also_default.json
python3 - gpurun_out/$TAG/bench_default.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = d['roofline']
print('default: value %.4g, ms_per_step %.5f, events %.2f us, frac %.3f, one launch %.2f us, %d bytes' % (d['value'], d['ms_per_step'], r['launch_us_events'], r['frac'], r['one_launch_us'] or 0, len(json.dumps(d, separators=(",", ":")))))
for w, row in r['configs'].items(): print('   %-40s %s' % (w, row))
PY
for a in "$@"; do
  timeout -k 10 200 python3 bench.py --no-also --no-cpu-baseline $a 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('$a:', 'events %.2f us' % r['launch_us_events'], 'kernel', r['kernel'], 'launches', r['launches_per_step'], 'valu frac %.3f' % (r['valu'] or {}).get('frac', 0))" || exit 1
done | tee gpurun_out/$TAG/bench_extra.log
