#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/r4_a
{
for lib in lib_s70_u_stamps lib_s100_u_stamps; do for integ in euler; do
  echo "== $lib $integ"
  ROBOY_SIM_LIB=$PWD/gym_roboy_amd/csrc/variants/$lib.so timeout -k 10 120 python tools/helper_stamps.py $integ 8192
done; done
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4_a/helpers_stamps_sweeps.log
