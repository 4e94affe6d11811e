#!/bin/bash
# round 4: the upper body's fused env layer (tree_lane_env_step / tree_split_env_step): goal rows fetched before (variant) / after the acceleration
cd /root/repo
mkdir -p gpurun_out/r4_a
{
for lib in gym_roboy_amd/csrc/variants/lib_early_goal.so gym_roboy_amd/csrc/libroboy_sim.so; do
  for integ in euler rk4; do
    echo "== $lib $integ"
    ROBOY_SIM_LIB=$PWD/$lib VECENV_ROBOT=upper VECENV_INTEGRATOR=$integ VECENV_SIZES=8192,65536,131072 timeout -k 10 200 python tools/vecenv_bench.py 2>&1 | grep fused
  done
done
} 2>&1 | tee gpurun_out/r4_a/envlane_goal.log
