#!/bin/bash
# the env-per-lane joint-tree form (arms as pair values) over the batch size, Euler and RK4; split form beside it where it applies
cd /root/repo
for integ in euler rk4; do for n in 8192 16384 32768 65536 131072 262144; do for k in 1 4; do
[ $k -eq 4 ] && [ $n -gt 32768 ] && continue
timeout -k 10 200 python bench.py --no-cpu-baseline --no-also --workload upper-body-8192-$integ --envs $n --kernel $k 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('upper body $integ envs $n kernel $k: %.2f us per step, %.3e env-steps/s' % (d['roofline']['launch_us_events'], d['value']))"
done; done; done
