#!/bin/bash
# round 4: the helpers' share of their parts' tendons (55 / 70 (library) / 85 / 100 %), and no helpers, on one box
cd /root/repo
mkdir -p gpurun_out/r4_a
run() { lib=$1; w=$2; n=$3
ROBOY_SIM_LIB=$PWD/$lib timeout -k 10 200 python bench.py --no-cpu-baseline --no-also --workload $w --envs $n 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('$lib $w envs $n: %.2f us per step (events)' % (d['roofline']['launch_us_events']))"
}
{
for lib in gym_roboy_amd/csrc/libroboy_sim.so gym_roboy_amd/csrc/variants/lib_s70_t1.so gym_roboy_amd/csrc/variants/lib_s45_u.so gym_roboy_amd/csrc/variants/lib_s70_u.so gym_roboy_amd/csrc/variants/lib_s85_u.so gym_roboy_amd/csrc/variants/lib_s100_u.so; do
 run $lib upper-body-8192-euler 8192; run $lib upper-body-8192-rk4 8192
done
} 2>&1 | tee gpurun_out/r4_a/helpers_sweeps.log
