#!/bin/bash
# the lean two-part split form of the joint-tree kernels (rb_kernel 6): parity, then us per step against the five-wave form (4) and one
# wave per 64 envs (1) between 16 384 and 49 152 envs; the fused env layer likewise
cd /root/repo
TAG=${1:-r5_a}
mkdir -p gpurun_out/$TAG
timeout -k 10 900 python -m pytest tests/test_tree_robot_gpu.py tests/test_full_size_gpu.py -x -q -m gpu -k "upper" > gpurun_out/$TAG/split2_tests.log 2>&1 || { tail -40 gpurun_out/$TAG/split2_tests.log; exit 1; }
tail -3 gpurun_out/$TAG/split2_tests.log
(for integ in euler rk4; do for n in 16384 20480 24576 32768 40960 49152; do for k in 6 1 4; do
  timeout -k 10 120 python3 bench.py --no-also --no-cpu-baseline --workload upper-body-8192-$integ --envs $n --kernel $k 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('$integ n=$n kernel $k: %.2f us per step (%s, %d launch(es) per step)' % (r['launch_us_events'], r['kernel'], r['launches_per_step']))" || exit 1
done; done; done
for k in 6 1 4; do VECENV_ROBOT=upper VECENV_KERNEL=$k VECENV_SIZES=20480,32768 timeout -k 10 200 python3 tools/vecenv_bench.py 2>/dev/null || exit 1; done) | tee gpurun_out/$TAG/split2_sweep.log
