#!/bin/bash
# round 4: fine sweep of the helpers' share of the arms' tendons (two sweeps; with / without the shared trunk), two passes over the
# libraries in gym_roboy_amd/csrc/variants/ and the shipped one, plain step only
cd /root/repo
mkdir -p gpurun_out/r4_a
run() { lib=$1; w=$2; n=$3
ROBOY_SIM_LIB=$PWD/$lib timeout -k 10 200 python bench.py --no-cpu-baseline --no-also --workload $w --envs $n 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('$(basename $1) $w: %.2f' % (d['roofline']['launch_us_events']))"
}
{
for pass in 1 2; do
for lib in gym_roboy_amd/csrc/libroboy_sim.so $(ls gym_roboy_amd/csrc/variants/v_*.so); do
 run $lib upper-body-8192-euler 8192; run $lib upper-body-8192-rk4 8192
done; done
} 2>&1 | tee gpurun_out/r4_a/share_sweep_fine.log
