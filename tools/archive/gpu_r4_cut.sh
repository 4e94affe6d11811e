#!/bin/bash
# round 4: A/B of split-form variants (libraries in gym_roboy_amd/csrc/variants/) on configs[3]: plain step and fused env step
cd /root/repo
mkdir -p gpurun_out/r4_a
LIBS=${LIBS:-"gym_roboy_amd/csrc/libroboy_sim.so"}
LOG=${LOG:-gpurun_out/r4_a/split_variants.log}
run() { lib=$1; w=$2; n=$3
ROBOY_SIM_LIB=$PWD/$lib timeout -k 10 200 python bench.py --no-cpu-baseline --no-also --workload $w --envs $n 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('$lib $w envs $n: %.2f us per step (events)' % (d['roofline']['launch_us_events']))"
}
{
for lib in $LIBS; do
 run $lib upper-body-8192-euler 8192; run $lib upper-body-8192-rk4 8192
 for integ in euler rk4; do
  ROBOY_SIM_LIB=$PWD/$lib VECENV_ROBOT=upper VECENV_SIZES=${SIZES:-8192} VECENV_INTEGRATOR=$integ timeout -k 10 200 python tools/vecenv_bench.py 2>/dev/null | sed "s#^#$lib #"
 done
done
} 2>&1 | tee $LOG
