#!/bin/bash
# Round 3, after the profile pass: whole GPU suite, the PPO update's kernel breakdown, rollout time of the training driver at world 1 / 2
set -o pipefail
cd /root/repo
OUT=/root/repo/gpurun_out/r3_a
mkdir -p $OUT
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -4 $OUT/pytest_gpu.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 120 python tools/train_rollout_time.py > $OUT/train_rollout_world1.json 2> $OUT/train_rollout_world1.err; echo "rollout world1 rc=$?"; cat $OUT/train_rollout_world1.json
timeout -k 10 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29701 tools/train_rollout_time.py 2> $OUT/train_rollout_world2.err | grep '^{' > $OUT/train_rollout_world2.json; echo "rollout world2 rc=$?"; cat $OUT/train_rollout_world2.json
export TMPDIR=/tmp
cd /tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ppo_prof -- python3 /root/repo/tools/ppo_update_profile.py > /dev/null 2> $OUT/ppo_prof.err; echo "ppo profile rc=$?"
f=$(find $OUT/ppo_prof -name "*kernel_stats.csv" | head -1)
python3 - "$f" > $OUT/ppo_update_kernel_stats.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("rocprofv3 --kernel-trace --stats -- python3 tools/ppo_update_profile.py   (three PPO iterations, 262 144 MsjRobot envs, fused kernels)")
for r in rows[:20]:
    print("%-100s calls %6s total_ms %9.2f avg_us %9.2f pct %5.1f" % (r["Name"][:100], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
PY
head -8 $OUT/ppo_update_kernel_stats.txt
