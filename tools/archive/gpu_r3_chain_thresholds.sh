#!/bin/bash
# one launch per step vs two chains (forced) around the thresholds, default bench regions
cd /root/repo
run() { w=$1; n=$2; k=$3; for c in 1 2; do
ROBOY_SIM_CHAINS=$c timeout -k 10 200 python bench.py --no-cpu-baseline --no-also --workload $w --envs $n --kernel $k 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('$w envs $n chains $c: %.2f us per step' % d['roofline']['launch_us_events'])"
done; }
run msj-262144-rk4 98304 0; run msj-262144-rk4 131072 0; run msj-262144-rk4 196608 0
run msj-262144-euler 131072 0; run msj-262144-euler 262144 0; run msj-262144-euler 393216 0; run msj-262144-euler 524288 0
run upper-body-8192-euler 32768 1; run upper-body-8192-euler 49152 1; run upper-body-8192-euler 65536 1
run upper-body-8192-rk4 49152 1; run upper-body-8192-rk4 65536 1; run upper-body-8192-rk4 131072 1
