#!/bin/bash
# round 4: tendon-helper waves in the split form of the joint-tree kernels: parity tests, then configs[3] timings
cd /root/repo
mkdir -p gpurun_out/r4_a
timeout -k 10 1000 python -m pytest tests/test_tree_robot_gpu.py tests/test_random_robots_gpu.py tests/test_full_size_gpu.py tests/test_env_layer_gpu.py -x -q -m gpu > gpurun_out/r4_a/helpers_tests.log 2>&1; rc=$?
tail -15 gpurun_out/r4_a/helpers_tests.log
[ $rc -ne 0 ] && exit $rc
run() { w=$1; n=$2; k=$3
timeout -k 10 200 python bench.py --no-cpu-baseline --no-also --workload $w --envs $n --kernel $k 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('$w envs $n kernel $k: %.2f us per step (events), %.2f wall, feasible %.4f' % (d['roofline']['launch_us_events'], d['ms_per_step']*1e3, d['sanity']['feasible_frac']))"
}
{
for n in 64 4096 8192 12288 16384; do run upper-body-8192-euler $n 0; done
for n in 8192 12288; do run upper-body-8192-rk4 $n 0; done
run upper-body-8192-euler 8192 1
VECENV_ROBOT=upper VECENV_SIZES=8192 timeout -k 10 200 python tools/vecenv_bench.py 2>&1 | grep fused
VECENV_ROBOT=upper VECENV_INTEGRATOR=rk4 VECENV_SIZES=8192 timeout -k 10 200 python tools/vecenv_bench.py 2>&1 | grep fused
} 2>&1 | tee gpurun_out/r4_a/helpers_sweep.log
