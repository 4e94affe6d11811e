#!/bin/bash
# Round 2 profile pass (one gpurun call): 2-rank rehearsal of the multi-GPU bench path (gloo, both ranks on
# the one GPU), rocprofv3 kernel stats of the default bench command, PMC HBM-traffic passes per workload,
# SQ counters of the headline and of the joint-tree kernel.  Outputs under gpurun_out/r2_a/.
set -o pipefail
cd /root/repo
OUT=/root/repo/gpurun_out/r2_a
mkdir -p $OUT
# --- 2 ranks on one GPU over gloo: the N > 1 code path with a real process group
ROBOY_BENCH_BACKEND=gloo timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29655 bench.py --gpus 2 --steps 20 --warmup 5 > $OUT/bench_2rank_gloo.json 2> $OUT/bench_2rank_gloo.err; echo "2-rank rehearsal rc=$?"
timeout -k 10 400 python bench.py > $OUT/bench_unprofiled.json 2> $OUT/bench_unprofiled.err; echo "bench rc=$?"
export TMPDIR=/tmp
cd /tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_stats -- python3 /root/repo/bench.py --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $OUT/prof_stats.err; echo "rocprof stats rc=$?"
for W in msj-262144-rk4 msj-262144-euler msj-2097152-euler msj-4096-euler upper-body-8192-euler upper-body-8192-rk4; do
 for C in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 200 rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_${W}_$C -- python3 /root/repo/bench.py --no-cpu-baseline --no-also --workload $W --steps 40 --warmup 8 --repeats 1 --no-graph > /dev/null 2> $OUT/pmc_${W}_$C.err; echo "pmc $W $C rc=$?"
 done
done
for C in FETCH_SIZE WRITE_SIZE; do
  VECENV_SIZES=2097152 timeout -k 10 200 rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_fused-env-2097152_$C -- python3 /root/repo/tools/vecenv_bench.py > /dev/null 2> $OUT/pmc_fused-env_$C.err; echo "pmc fused-env $C rc=$?"
done
for W in msj-262144-rk4 msj-2097152-euler upper-body-8192-euler; do
  timeout -k 10 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $OUT/pmc_${W}_SQ1 -- python3 /root/repo/bench.py --no-cpu-baseline --no-also --workload $W --steps 40 --warmup 8 --repeats 1 --no-graph > /dev/null 2> $OUT/pmc_${W}_SQ1.err; echo "pmc $W SQ1 rc=$?"
  timeout -k 10 200 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_${W}_SQ2 -- python3 /root/repo/bench.py --no-cpu-baseline --no-also --workload $W --steps 40 --warmup 8 --repeats 1 --no-graph > /dev/null 2> $OUT/pmc_${W}_SQ2.err; echo "pmc $W SQ2 rc=$?"
done
find $OUT -name "*.csv" | wc -l
