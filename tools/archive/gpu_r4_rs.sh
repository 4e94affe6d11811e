#!/bin/bash
# round 4: the rolled-stage env-per-lane RK4 form (ROBOY_SIM_RS=1 pinned tendons / 2 free scheduling) against the first form (0)
cd /root/repo
mkdir -p gpurun_out/r4_a
run() { w=$1; n=$2; rs=$3; c=$4; st=$5
ROBOY_SIM_RS=$rs ROBOY_SIM_CHAINS=$c timeout -k 10 200 python bench.py --no-cpu-baseline --no-also --workload $w --envs $n --kernel 1 $st 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('$w envs $n rs $rs chains $c $st: %.2f us per step (events), %.2f wall, feasible %.4f' % (d['roofline']['launch_us_events'], d['ms_per_step']*1e3, d['sanity']['feasible_frac']))"
}
{
for n in 131072 262144 2097152; do for rs in 0 1 2; do for c in 1 2; do run msj-262144-rk4 $n $rs $c; done; done; done
for rs in 0 1; do for c in 1 2; do run msj-262144-rk4 262144 $rs $c "--steps 20"; done; done
for n in 262144 2097152; do for rs in 0 3; do for c in 1 2; do run msj-262144-euler $n $rs $c; done; done; done
} 2>&1 | tee gpurun_out/r4_a/rs_sweep.log
