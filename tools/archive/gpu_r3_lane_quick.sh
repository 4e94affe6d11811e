#!/bin/bash
# Round 3: quick check of the env-per-lane joint-tree kernels after a generator change: tree parity tests, timing at three batch sizes
set -o pipefail
cd /root/repo
OUT=/root/repo/gpurun_out
mkdir -p $OUT
timeout -k 10 900 python -m pytest tests/test_tree_robot_gpu.py tests/test_full_size_gpu.py -m gpu -x -q > $OUT/pytest_lane.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -5 $OUT/pytest_lane.log
[ $rc -eq 0 ] || exit $rc
for w in upper-body-8192-euler upper-body-8192-rk4; do
  for n in 64 8192 16384 65536 262144; do
      timeout -k 10 200 python bench.py --workload $w --envs $n --kernel ${KERNEL:-1} --no-also --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$w n=$n', 'launch_us', round(d['roofline']['launch_us_events'],2), 'value', '%.3e'%d['value'])" | tee -a $OUT/lane_quick.log
  done
done
