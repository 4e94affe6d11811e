#!/bin/bash
# joint-tree lane kernel: one launch per step vs two chains (forced), Euler and RK4, over the batch size
cd /root/repo
for integ in euler rk4; do for n in 65536 131072 262144; do for c in 1 2; do
ROBOY_SIM_CHAINS=$c timeout -k 10 200 python bench.py --no-cpu-baseline --no-also --workload upper-body-8192-$integ --envs $n --kernel 1 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('upper body $integ envs $n chains $c (%s): %.2f us per step' % (d['roofline']['launches_per_step'], d['roofline']['launch_us_events']))"
done; done; done
