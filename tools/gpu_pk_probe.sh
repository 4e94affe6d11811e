#!/bin/bash
# Round 6: the price of the one-wave joint-tree kernel's non-arithmetic issue slots (tools/pk_const_probe.hip; built here by
# `hipcc --offload-arch=gfx950 -O3 -o tools/bin/pk_const_probe tools/pk_const_probe.hip`, the binary travels with the snapshot)
#   gpurun -- ./tools/gpu_pk_probe.sh <tag>   -> gpurun_out/<tag>/pk_const_probe.log
cd /root/repo
OUT=gpurun_out/${1:-r6_pk}
mkdir -p $OUT
[ -x tools/bin/pk_const_probe ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -Wno-unused-result -o tools/bin/pk_const_probe tools/pk_const_probe.hip || exit 1
timeout -k 10 120 ./tools/bin/pk_const_probe | tee $OUT/pk_const_probe.log
