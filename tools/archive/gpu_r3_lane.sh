#!/bin/bash
# Round 3: env-per-lane joint-tree kernels (tree_lane.hpp): parity tests, then lane (kernel 1) vs octets (kernel 3)
# over the batch size, Euler and RK4
set -o pipefail
cd /root/repo
OUT=/root/repo/gpurun_out
mkdir -p $OUT
timeout -k 10 900 python -m pytest tests/test_tree_robot_gpu.py tests/test_full_size_gpu.py tests/test_random_robots_gpu.py tests/test_env_layer_gpu.py -m gpu -x -q > $OUT/pytest_lane.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -15 $OUT/pytest_lane.log
[ $rc -eq 0 ] || exit $rc
: > $OUT/lane_sweep.log
for w in upper-body-8192-euler upper-body-8192-rk4; do
  for n in 64 2048 8192 16384 65536 262144; do
    for k in 1 3; do
      timeout -k 10 200 python bench.py --workload $w --envs $n --kernel $k --no-also --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$w n=$n kernel=$k', 'launch_us', round(d['roofline']['launch_us_events'],2), 'value', '%.3e'%d['value'], d['sanity'])" | tee -a $OUT/lane_sweep.log
    done
  done
done
