#!/bin/bash
#   gpurun -- ./tools/gpu_call_overhead.sh <tag>   -> gpurun_out/<tag>/rollout_call_overhead.log (two passes)
cd /root/repo
OUT=gpurun_out/${1:-r6_co}
mkdir -p $OUT
for PASS in 1 2; do timeout -k 10 300 python3 tools/proto/rollout_call_overhead.py 60 2>&1 | sed "s/^/pass $PASS: /" | tee -a $OUT/rollout_call_overhead.log; done
