#!/bin/bash
# Round 6: pair constants of the generated joint-tree kernels as packed operands (two s_mov_b32 + v_pk_*) against per-half plain
# instructions with literal operands (-DRBL_K2_SPLIT=1: gym_roboy_amd/csrc/variants/lib_k2split.so) - parity first, then us per step.
#   gpurun -- ./tools/gpu_k2split_ab.sh <tag>
cd /root/repo
OUT=gpurun_out/${1:-r6_k2}
mkdir -p $OUT
V=$PWD/gym_roboy_amd/csrc/variants/lib_k2split.so
ROBOY_SIM_LIB=$V timeout -k 10 600 python -m pytest tests/test_tree_robot_gpu.py tests/test_full_size_gpu.py tests/test_dispatch_table.py -x -q -k "(upper or tree or body) and not jit and not msj_variant and not hiprtc and not single_pass and not multi_pass and not wave" > $OUT/parity_k2split.log 2>&1; rc=$?
tail -3 $OUT/parity_k2split.log
[ $rc -ne 0 ] && exit $rc
for PASS in 1 2; do
  BASE=1 WL="upper-body-8192-euler upper-body-8192-rk4 upper-body-65536-euler" ./tools/ab.sh | sed "s/^/pass $PASS: /" | tee -a $OUT/k2split_ab.log
  BASE=1 WL="upper-body-65536-euler" EXTRA="--envs 32768" ./tools/ab.sh | sed "s/^/pass $PASS: /" | tee -a $OUT/k2split_ab.log
  for LIB in gym_roboy_amd/csrc/libroboy_sim.so gym_roboy_amd/csrc/variants/lib_k2split.so; do
    for INTEG in euler rk4; do
      ROBOY_SIM_LIB=$PWD/$LIB VECENV_ROBOT=upper VECENV_INTEGRATOR=$INTEG VECENV_SIZES=8192,32768,65536 timeout -k 10 200 python3 tools/vecenv_bench.py 2>/dev/null | sed "s|^|pass $PASS: $(basename $LIB) |" | tee -a $OUT/k2split_ab.log
    done
  done
done
