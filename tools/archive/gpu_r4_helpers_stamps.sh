#!/bin/bash
# round 4: barrier stamps of the split kernels (a -DRB_SPLIT_STAMPS build in gym_roboy_amd/csrc/variants/): plain step and fused env step
cd /root/repo
mkdir -p gpurun_out/r4_a
LIB=${LIB:-lib_s80t_stamps}
{
for mode in step env; do for integ in ${INTEGS:-euler}; do
  echo "== $LIB $integ $mode"
  ROBOY_SIM_LIB=$PWD/gym_roboy_amd/csrc/variants/$LIB.so timeout -k 10 120 python tools/helper_stamps.py $integ 8192 $mode
done; done
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4_a/helpers_stamps_env.log
