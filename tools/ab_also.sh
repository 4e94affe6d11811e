#!/bin/bash
# A/B of library builds over the whole default bench line (headline + `also`): each variant in its own process
cd /root/repo
for f in gym_roboy_amd/csrc/variants/lib_*.so; do
  ROBOY_SIM_LIB=$PWD/$f timeout -k 5 400 python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().split('\n')[-1])
print('$(basename $f)', 'HEAD', round(d['roofline']['launch_us_events'],2))
for a in d.get('also', []):
    print('$(basename $f)', a.get('workload'), a.get('launch_us_events') and round(a['launch_us_events'],2), '%.3e' % a['value'])"
done
