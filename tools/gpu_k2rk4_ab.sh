#!/bin/bash
# Round 6: RK4 instances of the five-wave split form with pair constants per half in the part waves only (lib_k2rk4_1) / in the helper
# waves only (lib_k2rk4_2) against the shipped library (packed in RK4), us per step, three passes.
#   gpurun -- ./tools/gpu_k2rk4_ab.sh <tag>
cd /root/repo
OUT=gpurun_out/${1:-r6_k2r}
mkdir -p $OUT
for PASS in 1 2 3; do
  for LIB in gym_roboy_amd/csrc/libroboy_sim.so gym_roboy_amd/csrc/variants/lib_k2rk4_1.so gym_roboy_amd/csrc/variants/lib_k2rk4_2.so; do
    for ARGS in "--workload upper-body-8192-rk4" "--workload upper-body-8192-rk4 --envs 16384"; do
      ROBOY_SIM_LIB=$PWD/$LIB timeout -k 5 120 python bench.py $ARGS --no-also --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('pass $PASS $(basename $LIB) $ARGS', 'us', round(d['roofline']['launch_us_events'],3), d['roofline']['kernel'])" | tee -a $OUT/k2split_rk4_ab.log
    done
    ROBOY_SIM_LIB=$PWD/$LIB VECENV_ROBOT=upper VECENV_INTEGRATOR=rk4 VECENV_SIZES=8192 timeout -k 10 200 python3 tools/vecenv_bench.py 2>/dev/null | sed "s|^|pass $PASS $(basename $LIB) |" | tee -a $OUT/k2split_rk4_ab.log
  done
done
