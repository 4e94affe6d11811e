#!/bin/bash
# Round 6: pair constants per half in the EULER instances of the split joint-tree forms (shipped: tree_lane_split.hpp, RBL_K2_SPLIT_EULER)
# against the library before the change (gym_roboy_amd/csrc/variants/lib_base.so, packed everywhere), us per step, three passes.
#   gpurun -- ./tools/gpu_k2euler_ab.sh <tag>
cd /root/repo
OUT=gpurun_out/${1:-r6_k2e}
mkdir -p $OUT
for PASS in 1 2 3; do
  for LIB in gym_roboy_amd/csrc/variants/lib_base.so gym_roboy_amd/csrc/libroboy_sim.so; do
    for ARGS in "--workload upper-body-8192-euler" "--workload upper-body-8192-rk4" "--workload upper-body-65536-euler --envs 32768" "--workload upper-body-65536-euler --envs 16384" "--workload upper-body-65536-euler --envs 4096"; do
      ROBOY_SIM_LIB=$PWD/$LIB timeout -k 5 120 python bench.py $ARGS --no-also --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('pass $PASS $(basename $LIB) $ARGS', 'us', round(d['roofline']['launch_us_events'],3), d['roofline']['kernel'])" | tee -a $OUT/k2split_euler_ab.log
    done
    ROBOY_SIM_LIB=$PWD/$LIB VECENV_ROBOT=upper VECENV_INTEGRATOR=euler VECENV_SIZES=4096,8192,16384,32768 timeout -k 10 200 python3 tools/vecenv_bench.py 2>/dev/null | sed "s|^|pass $PASS $(basename $LIB) |" | tee -a $OUT/k2split_euler_ab.log
  done
done
