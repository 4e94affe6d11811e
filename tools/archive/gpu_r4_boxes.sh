#!/bin/bash
# round 4: the driver's bench command (headline only) on one box - called several times, each call lands on a fresh box: the spread of the
# 20-step number between boxes (appends to gpurun_out/r4_a/boxes.log)
cd /root/repo
mkdir -p gpurun_out/r4_a
for rep in 1 2 3; do
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-also --no-cpu-baseline 2>/dev/null | python -c "
import sys,json,socket
d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); r=d['roofline']
print('%s steps 20: value %.4g, wall %.3f us per step, events %.2f us, frac %.3f; one launch %.2f us' % (socket.gethostname(), d['value'], 1e3*d['ms_per_step'], r['launch_us_events'], r['frac'], r.get('one_launch_us') or 0))"
done | tee -a gpurun_out/r4_a/boxes.log
timeout -k 10 300 python bench.py --no-also --no-cpu-baseline 2>/dev/null | python -c "
import sys,json,socket
d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); r=d['roofline']
print('%s steps 400: value %.4g, wall %.3f us per step, events %.2f us, frac %.3f' % (socket.gethostname(), d['value'], 1e3*d['ms_per_step'], r['launch_us_events'], r['frac']))" | tee -a gpurun_out/r4_a/boxes.log
