#!/bin/bash
# A/B of kernel builds on the GPU box: every gym_roboy_amd/csrc/variants/lib_*.so (git-ignored), each in its own process, through
# bench.py (ROBOY_SIM_LIB); WL = workloads, EXTRA = further bench.py arguments (e.g. "--kernel 4 --envs 16384"), BASE=1 adds the shipped library
cd /root/repo
WL=${WL:-"msj-2097152-euler msj-262144-rk4 msj-4096-euler"}
LIBS=$(ls gym_roboy_amd/csrc/variants/lib_*.so 2>/dev/null)
[ -n "$BASE" ] && LIBS="gym_roboy_amd/csrc/libroboy_sim.so $LIBS"
for f in $LIBS; do
  for w in $WL; do
    ROBOY_SIM_LIB=$PWD/$f timeout -k 5 120 python bench.py --workload $w --no-also --no-cpu-baseline $EXTRA 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$(basename $f)', '$w', '$EXTRA', 'launch_us', round(d['roofline']['launch_us_events'],2), 'value', '%.3e'%d['value'], 'frac', round(d['roofline']['frac'],4), d['roofline']['kernel'])"
  done
done
