#!/bin/bash
# round 4: the split form with (library) and without (variants/lib_h0.so) tendon-helper waves, same box; barrier stamps of the latter
cd /root/repo
mkdir -p gpurun_out/r4_a
run() { lib=$1; w=$2; n=$3
ROBOY_SIM_LIB=$PWD/$lib timeout -k 10 200 python bench.py --no-cpu-baseline --no-also --workload $w --envs $n 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('$lib $w envs $n: %.2f us per step (events)' % (d['roofline']['launch_us_events']))"
}
{
for lib in gym_roboy_amd/csrc/variants/lib_h0.so gym_roboy_amd/csrc/libroboy_sim.so; do
 for n in 64 8192; do run $lib upper-body-8192-euler $n; run $lib upper-body-8192-rk4 $n; done
 ROBOY_SIM_LIB=$PWD/$lib VECENV_ROBOT=upper VECENV_SIZES=8192 timeout -k 10 200 python tools/vecenv_bench.py 2>&1 | grep fused
done
for integ in euler rk4; do echo "== stamps h0 $integ"; ROBOY_SIM_LIB=$PWD/gym_roboy_amd/csrc/variants/lib_h0_stamps.so timeout -k 10 120 python tools/helper_stamps.py $integ 8192 2>&1 | grep wave; done
} 2>&1 | tee gpurun_out/r4_a/helpers_ab.log
