#!/bin/bash
# batch-size sweep of both kernel forms (per-step time with hipGraph replay)
cd /root/repo
for n in 1024 4096 8192 16384 32768 65536 131072; do
  for k in 1 2; do
    for integ in msj-4096-euler msj-262144-rk4; do
      timeout -k 5 120 python bench.py --workload $integ --envs $n --kernel $k --steps 2000 --warmup 100 --no-also --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('n=$n kernel=$k $integ', 'us/step', round(d['roofline']['launch_us_events'],2), 'env-steps/s', '%.3e'%d['value'])"
    done
  done
done
