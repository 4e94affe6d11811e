#!/bin/bash
# phase stamps of the one-wave joint-tree kernels (needs gym_roboy_amd/csrc/variants/lib_lane_stamps.so: a -DRB_LANE_STAMPS build)
cd /root/repo
TAG=${1:-r5_a}
mkdir -p gpurun_out/$TAG
export ROBOY_SIM_LIB=$PWD/gym_roboy_amd/csrc/variants/lib_lane_stamps.so
for cfg in "euler 65536" "euler 65536 env" "euler 131072" "rk4 65536"; do
  timeout -k 10 120 python3 tools/lane_stamps.py $cfg || exit 1
done > gpurun_out/$TAG/lane_stamps.log 2>&1
cat gpurun_out/$TAG/lane_stamps.log
