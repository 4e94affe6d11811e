#!/bin/bash
# PMC passes for the headline workload (4 096 envs, Euler) and configs[2] (262 144, RK4)
set -o pipefail
OUT=/root/repo/gpurun_out
export TMPDIR=/tmp
cd /tmp
for W in msj-4096-euler msj-262144-rk4; do
 for C in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 200 rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_${W}_$C -- python3 /root/repo/bench.py --no-cpu-baseline --no-also --workload $W --steps 200 --warmup 20 --no-graph > /dev/null 2> $OUT/pmc_${W}_$C.err; echo "pmc $W $C rc=$?"
 done
done
