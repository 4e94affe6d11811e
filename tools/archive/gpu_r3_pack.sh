#!/bin/bash
# packed-mates lane kernels: joint-tree tests, then both forms' step times
set -o pipefail
cd /root/repo
timeout -k 10 600 python -m pytest tests/test_tree_robot_gpu.py tests/test_full_size_gpu.py -x -q -m gpu 2>&1 | tail -4 || exit 1
for w in upper-body-8192-euler upper-body-8192-rk4 upper-body-65536-euler; do for k in 1 4; do
timeout -k 10 200 python bench.py --no-cpu-baseline --no-also --workload $w --kernel $k 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('$w kernel $k', round(d['roofline']['launch_us_events'],2), '%.3e' % d['value'])"
done; done
