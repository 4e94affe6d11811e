#!/bin/bash
# round 4: the cut form of the split joint-tree kernels (heavy parts as a proximal and a distal wave) - parity tests on the variant
# library, then configs[3] timings beside the helper form
cd /root/repo
mkdir -p gpurun_out/r4_a
LIBS=${LIBS:-"gym_roboy_amd/csrc/variants/lib_s70_u.so gym_roboy_amd/csrc/variants/lib_cut0.so"}
TESTLIB=${TESTLIB:-gym_roboy_amd/csrc/variants/lib_cut0.so}
run() { lib=$1; w=$2; n=$3
ROBOY_SIM_LIB=$PWD/$lib timeout -k 10 200 python bench.py --no-cpu-baseline --no-also --workload $w --envs $n 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('$lib $w envs $n: %.2f us per step (events)' % (d['roofline']['launch_us_events']))"
}
{
if [ -n "$TESTLIB" ]; then
ROBOY_SIM_LIB=$PWD/$TESTLIB timeout -k 10 600 python -m pytest tests/test_tree_robot_gpu.py tests/test_random_robots_gpu.py -m gpu -x -q 2>&1 | tail -5 || exit 1
fi
for lib in $LIBS; do
 run $lib upper-body-8192-euler 8192; run $lib upper-body-8192-rk4 8192
done
} 2>&1 | tee gpurun_out/r4_a/cut_form.log
