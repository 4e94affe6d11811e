#!/bin/bash
# What a short rollout call costs, on rocprofv3's clocks (tools/proto/rollout_call_timeline.py): K = 20 (the driver's) and K = 400.
#   gpurun -- ./tools/gpu_call_timeline.sh <tag>    -> gpurun_out/<tag>/call_timeline.log
OUT=/root/repo/gpurun_out/${1:-r6_tl}
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for K in 20 400; do
  CALLS=40; [ $K = 400 ] && CALLS=8
  timeout -k 10 300 rocprofv3 --kernel-trace --hip-trace --output-format csv -d $OUT/trace_k$K -- python3 /root/repo/tools/proto/rollout_call_timeline.py $K $CALLS > $OUT/trace_k$K.out 2> $OUT/trace_k$K.err; echo "trace K=$K rc=$?"
  python3 /root/repo/tools/proto/rollout_call_timeline.py --analyse $OUT/trace_k$K | tee -a $OUT/call_timeline.log
done
# the raw traces are large: keep the summaries
rm -rf $OUT/trace_k20 $OUT/trace_k400
