#!/bin/bash
# number of chains (forced) over workloads: upper body lane kernel and MSJ
cd /root/repo
run() { w=$1; n=$2; k=$3; for c in 1 2 3 4; do
ROBOY_SIM_CHAINS=$c timeout -k 10 200 python bench.py --no-cpu-baseline --no-also --workload $w --envs $n --kernel $k 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('$w envs $n chains $c (%s): %.2f us per step' % (d['roofline']['launches_per_step'], d['roofline']['launch_us_events']))"
done; }
run upper-body-8192-euler 131072 1
run upper-body-8192-euler 262144 1
run upper-body-8192-rk4 262144 1
run msj-262144-rk4 524288 0
run msj-262144-rk4 2097152 0
run msj-262144-euler 1048576 0
run msj-262144-euler 2097152 0
