#!/bin/bash
mkdir -p gpurun_out/r4_a tools/bin
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fno-slp-vectorize -o tools/bin/region_timeline_probe tools/proto/region_timeline_probe.hip || exit 1
timeout -k 10 120 tools/bin/region_timeline_probe | tee gpurun_out/r4_a/region_timeline.log
