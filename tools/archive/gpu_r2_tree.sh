#!/bin/bash
# Round 2: joint-tree kernel (tree_aba.hpp): parity tests, then timing of configs[3]
set -o pipefail
cd /root/repo
OUT=/root/repo/gpurun_out
mkdir -p $OUT
timeout -k 10 600 python -m pytest tests/test_tree_robot_gpu.py tests/test_full_size_gpu.py tests/test_env_layer_gpu.py tests/test_ppo.py -m gpu -x -q > $OUT/pytest_tree.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -15 $OUT/pytest_tree.log
[ $rc -eq 0 ] || exit $rc
for w in upper-body-8192-euler upper-body-8192-rk4; do
  timeout -k 10 200 python bench.py --workload $w --no-also --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$w', 'launch_us', round(d['roofline']['launch_us_events'],2), 'value', '%.3e'%d['value'], d['sanity'])"
done
