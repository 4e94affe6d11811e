#!/bin/bash
# joint-tree kernel check after a change: parity tests, then the two upper-body timings
set -o pipefail
cd /root/repo
timeout -k 10 600 python -m pytest tests/test_random_robots_gpu.py tests/test_tree_robot_gpu.py tests/test_full_size_gpu.py tests/test_env_layer_gpu.py -x -q -m gpu 2>&1 | tail -5 || exit 1
for w in upper-body-8192-euler upper-body-8192-rk4; do
  timeout -k 10 200 python bench.py --workload $w --no-also --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$w', 'launch_us', round(d['roofline']['launch_us_events'],2), 'value', '%.3e'%d['value'])" || exit 1
done
