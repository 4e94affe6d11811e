#!/bin/bash
# A round's profile pass (one gpurun call; raw output gpurun_out/<raw tag>/, then `python tools/summarize_profile.py <raw tag> <dir
# under profiles/>`): 2-rank rehearsal over gloo, the driver's bench command unprofiled, rocprofv3 --kernel-trace --stats of the SAME
# command, separate --pmc FETCH_SIZE / WRITE_SIZE passes (no tracing) and SQ counter passes of the workloads named below.
#   gpurun --timeout 1200 -- ./tools/gpu_profile_round.sh r6_p
set -o pipefail
cd /root/repo
OUT=/root/repo/gpurun_out/${1:-r6_p}
mkdir -p $OUT
ROBOY_BENCH_BACKEND=gloo timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29655 bench.py --gpus 2 --steps 20 --warmup 5 > $OUT/bench_2rank_gloo.json 2> $OUT/bench_2rank_gloo.err; echo "2-rank rehearsal rc=$?"
timeout -k 10 500 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_unprofiled.json 2> $OUT/bench_unprofiled.err; echo "bench rc=$?"
cp bench_also.json $OUT/bench_also_unprofiled.json
export TMPDIR=/tmp
cd /tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_stats -- python3 /root/repo/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $OUT/prof_stats.err; echo "rocprof stats rc=$?"
pmc() { W=$1; TAG=$2; shift 2
 for C in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 200 rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_${TAG}_$C -- python3 /root/repo/bench.py --no-cpu-baseline --no-also --workload $W --steps 40 --warmup 8 --repeats 1 --no-graph "$@" > /dev/null 2> $OUT/pmc_${TAG}_$C.err; echo "pmc $TAG $C rc=$?"
 done
}
sq() { W=$1; TAG=$2; shift 2;
  timeout -k 10 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $OUT/pmc_${TAG}_SQ1 -- python3 /root/repo/bench.py --no-cpu-baseline --no-also --workload $W --steps 40 --warmup 8 --repeats 1 --no-graph "$@" > /dev/null 2> $OUT/pmc_${TAG}_SQ1.err; echo "pmc $TAG SQ1 rc=$?"
  timeout -k 10 200 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_${TAG}_SQ2 -- python3 /root/repo/bench.py --no-cpu-baseline --no-also --workload $W --steps 40 --warmup 8 --repeats 1 --no-graph "$@" > /dev/null 2> $OUT/pmc_${TAG}_SQ2.err; echo "pmc $TAG SQ2 rc=$?"
}
# the hash of the kernel sources these counters belong to (bench.py: csrc_hash; the summariser stamps every row with it)
python3 -c "import sys; sys.path.insert(0, '/root/repo'); import bench; print(bench.csrc_hash())" > $OUT/csrc_hash.txt; echo "csrc hash $(cat $OUT/csrc_hash.txt)"
# every workload whose traffic bench.py quotes: the headline, the rows of roofline.configs, the fused env rows (eager, one launch per step)
pmc msj-262144-rk4 msj-262144-rk4
pmc msj-262144-euler msj-262144-euler
pmc msj-2097152-euler msj-2097152-euler
pmc msj-4096-euler msj-4096-euler
pmc upper-body-8192-euler upper-body-8192-euler
pmc upper-body-8192-rk4 upper-body-8192-rk4
pmc upper-body-65536-euler upper-body-65536-euler --kernel 1
fused() { NAME=$1; shift
 for C in FETCH_SIZE WRITE_SIZE; do
  env "$@" timeout -k 10 200 rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_${NAME}_$C -- python3 /root/repo/tools/vecenv_bench.py > /dev/null 2> $OUT/pmc_${NAME}_$C.err; echo "pmc $NAME $C rc=$?"
 done
}
# (`env VAR=... timeout ... rocprofv3 -- python3 ...`: env and timeout run BEFORE the profiler starts; the profiled program itself is python3)
fused fused-env-2097152 VECENV_SIZES=2097152
fused fused-env-UpperBodyRobot-8192 VECENV_ROBOT=upper VECENV_SIZES=8192
fused fused-env-UpperBodyRobot-65536 VECENV_ROBOT=upper VECENV_SIZES=65536
fused fused-env-32768-rk4-k5 VECENV_KERNEL=5 VECENV_INTEGRATOR=rk4 VECENV_SIZES=32768
sq msj-262144-rk4 msj-262144-rk4
sq upper-body-8192-euler upper-body-8192-euler
sq upper-body-8192-rk4 upper-body-8192-rk4
sq upper-body-65536-euler upper-body-65536-euler --kernel 1
find $OUT -name "*.csv" | wc -l
