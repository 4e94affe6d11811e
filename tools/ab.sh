#!/bin/bash
# A/B kernel builds on the GPU box: each variant in its own process, same workloads.
cd /root/repo
WL=${WL:-"msj-2097152-euler msj-262144-rk4 msj-4096-euler"}
for f in gym_roboy_amd/csrc/variants/lib_*.so; do
  for w in $WL; do
    ROBOY_SIM_LIB=$PWD/$f timeout -k 5 120 python bench.py --workload $w --no-also --no-cpu-baseline $EXTRA 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$(basename $f)', '$w', 'launch_us', round(d['roofline']['launch_us_events'],2), 'value', '%.3e'%d['value'], 'frac', round(d['roofline']['frac'],4))"
  done
done
