#!/bin/bash
# Round 3 profile pass (one gpurun call): default bench line, 2-rank rehearsal (gloo, both ranks on the one GPU), rocprofv3
# kernel stats of the default bench command, PMC HBM-traffic passes per workload, SQ counters of the headline kernel
# (262 144 and 2 097 152 envs) and of the joint-tree kernels (split, env-per-lane and octets), rollout time of the training
# driver at world 1 / 2 (ranks measuring in turns), the PPO update's kernel breakdown.  Outputs under gpurun_out/r3_a/; tools/summarize_profile.py r3_a turns them into profiles/r3_a/.
set -o pipefail
cd /root/repo
OUT=/root/repo/gpurun_out/r3_a
mkdir -p $OUT
ROBOY_BENCH_BACKEND=gloo timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29655 bench.py --gpus 2 --steps 20 --warmup 5 > $OUT/bench_2rank_gloo.json 2> $OUT/bench_2rank_gloo.err; echo "2-rank rehearsal rc=$?"
timeout -k 10 500 python bench.py > $OUT/bench_unprofiled.json 2> $OUT/bench_unprofiled.err; echo "bench rc=$?"
timeout -k 10 120 python tools/train_rollout_time.py > $OUT/train_rollout_world1.json 2> $OUT/train_rollout_world1.err; echo "rollout world1 rc=$?"
timeout -k 10 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29701 tools/train_rollout_time.py 2> $OUT/train_rollout_world2.err | grep '^{' > $OUT/train_rollout_world2.json; echo "rollout world2 rc=$?"
export TMPDIR=/tmp
cd /tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_stats -- python3 /root/repo/bench.py --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $OUT/prof_stats.err; echo "rocprof stats rc=$?"
for W in msj-262144-rk4 msj-262144-euler msj-2097152-euler msj-2097152-rk4 msj-4096-euler upper-body-8192-euler upper-body-8192-rk4 upper-body-65536-euler; do
 for C in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 200 rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_${W}_$C -- python3 /root/repo/bench.py --no-cpu-baseline --no-also --workload $W --steps 40 --warmup 8 --repeats 1 --no-graph > /dev/null 2> $OUT/pmc_${W}_$C.err; echo "pmc $W $C rc=$?"
 done
done
for C in FETCH_SIZE WRITE_SIZE; do
  VECENV_SIZES=2097152 timeout -k 10 200 rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_fused-env-2097152_$C -- python3 /root/repo/tools/vecenv_bench.py > /dev/null 2> $OUT/pmc_fused-env_$C.err; echo "pmc fused-env $C rc=$?"
  VECENV_ROBOT=upper VECENV_SIZES=65536 timeout -k 10 200 rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_fused-env-UpperBodyRobot-65536_$C -- python3 /root/repo/tools/vecenv_bench.py > /dev/null 2> $OUT/pmc_fused-env-upper_$C.err; echo "pmc fused-env upper $C rc=$?"
  VECENV_ROBOT=upper VECENV_SIZES=8192 timeout -k 10 200 rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_fused-env-UpperBodyRobot-8192_$C -- python3 /root/repo/tools/vecenv_bench.py > /dev/null 2> $OUT/pmc_fused-env-upper8k_$C.err; echo "pmc fused-env upper 8192 $C rc=$?"
done
sq() { W=$1; TAG=$2; shift 2; 
  timeout -k 10 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $OUT/pmc_${TAG}_SQ1 -- python3 /root/repo/bench.py --no-cpu-baseline --no-also --workload $W --steps 40 --warmup 8 --repeats 1 --no-graph "$@" > /dev/null 2> $OUT/pmc_${TAG}_SQ1.err; echo "pmc $TAG SQ1 rc=$?"
  timeout -k 10 200 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_${TAG}_SQ2 -- python3 /root/repo/bench.py --no-cpu-baseline --no-also --workload $W --steps 40 --warmup 8 --repeats 1 --no-graph "$@" > /dev/null 2> $OUT/pmc_${TAG}_SQ2.err; echo "pmc $TAG SQ2 rc=$?"
}
sq msj-262144-rk4 msj-262144-rk4
sq msj-2097152-rk4 msj-2097152-rk4
sq msj-2097152-euler msj-2097152-euler
sq upper-body-8192-euler upper-body-8192-euler
sq upper-body-8192-rk4 upper-body-8192-rk4
sq upper-body-65536-euler upper-body-65536-euler
sq upper-body-8192-euler upper-body-8192-euler-lane --kernel 1
sq upper-body-8192-rk4 upper-body-8192-rk4-lane --kernel 1
sq upper-body-8192-euler upper-body-8192-euler-octets --kernel 3
sq upper-body-8192-rk4 upper-body-8192-rk4-octets --kernel 3
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
find $OUT -name "*.csv" | wc -l
