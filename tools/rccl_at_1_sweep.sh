#!/bin/bash
# One-GPU rehearsal of the multi-rank bench path: RANK/WORLD_SIZE=1 + ROBOY_BENCH_DIST_AT_1=1 make
# bench.py bring up RCCL and run the statistics all-reduce, barrier and max-reduce with one rank.
cd /root/repo
show() { python -c "
import sys,json
lines=open('$1').read().splitlines()
assert len(lines) == 1, 'stdout must be exactly one line, got %d' % len(lines)
d=json.loads(lines[0]); print('$2', 'us/step', round(d['ms_per_step']*1e3,3), 'events', round(d['roofline']['launch_us_events'],3), 'value %.3e' % d['value'], 'stats', d['sanity']['allreduced_stats'][6])"; }
A="bench.py --gpus 1 --steps 3200 --warmup 100 --no-also --no-cpu-baseline"
o=gpurun_out/rccl_sweep; mkdir -p $o
timeout -k 10 200 python $A > $o/plain.out 2> $o/plain.err; show $o/plain.out no_process_group
export RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 ROBOY_BENCH_DIST_AT_1=1
p=29570
for se in 100 400 800; do
  p=$((p+1)); MASTER_PORT=$p ROBOY_BENCH_STATS_EVERY=$se timeout -k 10 200 python $A > $o/se$se.out 2> $o/se$se.err; show $o/se$se.out rccl_one_rank_stats_every_$se
done
unset RANK WORLD_SIZE LOCAL_RANK MASTER_ADDR
ROBOY_BENCH_DIST_AT_1=1 timeout -k 10 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29579 $A > $o/tr.out 2> $o/tr.err; show $o/tr.out torchrun_rccl_one_rank_default_interval
