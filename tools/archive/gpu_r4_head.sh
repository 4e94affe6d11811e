#!/bin/bash
# round 4: eager head in front of the rollout graphs (ROBOY_SIM_EAGER_HEAD=0/1), 20-step and default regions; chains tests first
cd /root/repo
mkdir -p gpurun_out/r4_a
timeout -k 10 900 python -m pytest tests/test_full_size_gpu.py tests/test_physics_gpu.py -x -q -m gpu > gpurun_out/r4_a/head_tests.log 2>&1 || { tail -40 gpurun_out/r4_a/head_tests.log; exit 1; }
tail -3 gpurun_out/r4_a/head_tests.log
run() { w=$1; n=$2; h=$3; c=$4; st=$5
ROBOY_SIM_EAGER_HEAD=$h ROBOY_SIM_CHAINS=$c timeout -k 10 200 python bench.py --no-cpu-baseline --no-also --workload $w --envs $n $st 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('$w envs $n head $h chains $c $st: %.2f us per step (events), %.2f wall' % (d['roofline']['launch_us_events'], d['ms_per_step']*1e3))"
}
{
for h in 0 1; do for c in 1 2; do run msj-262144-rk4 262144 $h $c "--steps 20"; done; done
for h in 0 1; do for c in 1 2; do run msj-262144-rk4 262144 $h $c "--steps 100"; done; done
for h in 0 1; do run msj-262144-rk4 262144 $h 2; done
for h in 0 1; do run msj-262144-rk4 131072 $h 2 "--steps 20"; done
for h in 0 1; do run msj-262144-euler 262144 $h 2 "--steps 20"; done
for h in 0 1; do run upper-body-65536-euler 65536 $h 2 "--steps 20"; done
} 2>&1 | tee gpurun_out/r4_a/head_sweep.log
