#!/bin/bash
# Fence scopes of a two-chain rollout's packets, read from ROCclr's log (AMD_LOG_LEVEL=4), with the chain events in their
# fence-free form (default) and with ROBOY_SIM_EVENT_SYSTEM_FENCE=1.  The raw logs are large; the packet / marker lines are kept.
#   gpurun -- ./tools/gpu_fence_probe.sh <tag>     -> gpurun_out/<tag>/chain_fence_{free,system}.log (+ .head: the first 400 raw lines)
cd /root/repo
OUT=gpurun_out/${1:-r6_fence}
mkdir -p $OUT
/opt/rocm/bin/hipconfig --version > $OUT/hip_version.txt 2>&1; cat /opt/rocm/.info/version >> $OUT/hip_version.txt 2>/dev/null
for MODE in free system; do
  if [ $MODE = system ]; then export ROBOY_SIM_EVENT_SYSTEM_FENCE=1; else unset ROBOY_SIM_EVENT_SYSTEM_FENCE; fi
  AMD_LOG_LEVEL=4 timeout -k 10 300 python3 tools/proto/chain_fence_probe.py > $OUT/probe_$MODE.out 2> /tmp/fence_$MODE.raw; echo "probe $MODE rc=$? raw lines $(wc -l < /tmp/fence_$MODE.raw)"
  # from the first marker on: packets, barriers, markers, the HIP calls that produce them
  awk '/=== PROBE A begin/{on=1} on' /tmp/fence_$MODE.raw | grep -E "PROBE|[Hh]eader|[Bb]arrier|acquire|release|[Ff]ence|hipEventRecord|hipStreamWaitEvent|hipGraphLaunch|hipLaunchKernel|hipModuleLaunch|ShaderName|Signal" > $OUT/chain_fence_$MODE.log || true
  awk '/=== PROBE A begin/{on=1} on' /tmp/fence_$MODE.raw | head -400 > $OUT/chain_fence_$MODE.head
  echo "kept $(wc -l < $OUT/chain_fence_$MODE.log) lines, $(du -k $OUT/chain_fence_$MODE.log | cut -f1) KiB"
done
unset ROBOY_SIM_EVENT_SYSTEM_FENCE
