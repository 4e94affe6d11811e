#!/bin/bash
# round 4: mid-size batches, RK4: env per lane (first form, one-wave workgroups) / rolled-stage form with one-wave workgroups / two lanes per env / tendon per lane
cd /root/repo
mkdir -p gpurun_out/r4_a
run() { w=$1; n=$2; k=$3; rs=$4
ROBOY_SIM_RS64=$rs ROBOY_SIM_CHAINS=1 timeout -k 10 200 python bench.py --no-cpu-baseline --no-also --workload $w --envs $n --kernel $k 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('$w envs $n kernel $k rs64 $rs: %.2f us per step (events), %.2f wall' % (d['roofline']['launch_us_events'], d['ms_per_step']*1e3))"
}
{
for n in 8192 16384 24576 32768 49152 65536; do run msj-262144-rk4 $n 1 0; run msj-262144-rk4 $n 1 1; run msj-262144-rk4 $n 5 0; run msj-262144-rk4 $n 2 0; done
for n in 4096 8192 16384 32768 65536; do run msj-262144-euler $n 1 0; run msj-262144-euler $n 5 0; run msj-262144-euler $n 2 0; done
} 2>&1 | tee gpurun_out/r4_a/mid_sweep.log
