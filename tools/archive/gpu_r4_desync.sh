#!/bin/bash
# round 4: fused env step with the episodes in lock-step (the default of the bench tools) and spread over an episode (training),
# before (variants/lib_two_draws.so) and after only the observable goal draw is evaluated on done + auto_reset
cd /root/repo
mkdir -p gpurun_out/r4_a
timeout -k 10 600 python -m pytest tests/test_env_layer_gpu.py tests/test_env_golden_gpu.py tests/test_tree_robot_gpu.py -m gpu -x -q 2>&1 | tail -3 || exit 1
{
for lib in gym_roboy_amd/csrc/variants/lib_two_draws.so gym_roboy_amd/csrc/libroboy_sim.so; do for ds in 0 1; do
  ROBOY_SIM_LIB=$PWD/$lib VECENV_DESYNC=$ds VECENV_ROBOT=upper VECENV_SIZES=8192,65536 timeout -k 10 200 python tools/vecenv_bench.py 2>/dev/null | sed "s#^#$(basename $lib) upper #"
  ROBOY_SIM_LIB=$PWD/$lib VECENV_DESYNC=$ds VECENV_ROBOT=msj VECENV_SIZES=262144,2097152 timeout -k 10 200 python tools/vecenv_bench.py 2>/dev/null | sed "s#^#$(basename $lib) msj #"
done; done
} 2>&1 | tee gpurun_out/r4_a/env_desync.log
