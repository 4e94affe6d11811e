#!/bin/bash
# round 4: the two-lanes-per-env form (kernel 5): parity tests, then per-step time against the env-per-lane form (kernel 1) and the
# tendon-per-lane form (kernel 2) over the batch size, one launch per step and two chains
cd /root/repo
mkdir -p gpurun_out/r4_a
timeout -k 10 600 python -m pytest tests/test_physics_gpu.py -x -q -m gpu -k "oracle or agree or pair" > gpurun_out/r4_a/pairs_tests.log 2>&1 || { tail -30 gpurun_out/r4_a/pairs_tests.log; exit 1; }
tail -3 gpurun_out/r4_a/pairs_tests.log
run() { w=$1; n=$2; k=$3; c=$4; st=$5
ROBOY_SIM_CHAINS=$c timeout -k 10 200 python bench.py --no-cpu-baseline --no-also --workload $w --envs $n --kernel $k $st 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('$w envs $n kernel $k chains $c $st: %.2f us per step (events), %.2f wall' % (d['roofline']['launch_us_events'], d['ms_per_step']*1e3))"
}
{
for n in 16384 32768 65536 131072 262144 524288 2097152; do for k in 1 5; do run msj-262144-rk4 $n $k 1; done; done
for n in 8192 16384; do run msj-262144-rk4 $n 2 1; done
for n in 131072 262144 524288 2097152; do for k in 1 5; do run msj-262144-rk4 $n $k 2; done; done
for k in 1 5; do for c in 1 2; do run msj-262144-rk4 262144 $k $c "--steps 20"; done; done
for n in 8192 16384 32768 65536 131072 262144 524288 2097152; do for k in 1 5; do run msj-262144-euler $n $k 1; done; done
for n in 262144 524288 2097152; do for k in 1 5; do run msj-262144-euler $n $k 2; done; done
} 2>&1 | tee gpurun_out/r4_a/pairs_sweep.log
