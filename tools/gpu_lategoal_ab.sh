#!/bin/bash
# Round 6: the fused env step of the one-wave joint-tree kernels with the goal rows fetched in front of the step (RBL_LATE_GOAL=0: the
# round-5 form, 56 bytes of scratch in the RK4 instance), behind it for RK4 only (2: shipped) or for both integrators (3).
#   gpurun -- ./tools/gpu_lategoal_ab.sh <tag>    (variants: gym_roboy_amd/csrc/variants/lib_lategoal{0,3}.so, built with -DRBL_LATE_GOAL=...)
cd /root/repo
OUT=gpurun_out/${1:-r6_lg}
mkdir -p $OUT
timeout -k 10 900 python -m pytest tests/test_code_objects.py tests/test_env_golden_gpu.py tests/test_env_layer_gpu.py tests/test_dispatch_table.py tests/test_tree_robot_gpu.py tests/test_random_robots_gpu.py -x -q > $OUT/parity.log 2>&1; rc=$?
tail -5 $OUT/parity.log
[ $rc -ne 0 ] && exit $rc
for LIB in gym_roboy_amd/csrc/variants/lib_lategoal0.so gym_roboy_amd/csrc/libroboy_sim.so gym_roboy_amd/csrc/variants/lib_lategoal3.so; do
  for PASS in 1 2; do
    for INTEG in euler rk4; do
      ROBOY_SIM_LIB=$PWD/$LIB VECENV_ROBOT=upper VECENV_INTEGRATOR=$INTEG VECENV_SIZES=65536,131072 VECENV_KERNEL=1 timeout -k 10 200 python3 tools/vecenv_bench.py 2>/dev/null | sed "s|^|$(basename $LIB) pass $PASS: |" | tee -a $OUT/lategoal_ab.log
      ROBOY_SIM_LIB=$PWD/$LIB VECENV_ROBOT=upper VECENV_INTEGRATOR=$INTEG VECENV_SIZES=65536 VECENV_KERNEL=1 VECENV_DESYNC=1 timeout -k 10 200 python3 tools/vecenv_bench.py 2>/dev/null | sed "s|^|$(basename $LIB) pass $PASS: |" | tee -a $OUT/lategoal_ab.log
    done
  done
done
