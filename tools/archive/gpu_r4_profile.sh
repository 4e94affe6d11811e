#!/bin/bash
# Round 4 profile pass (one gpurun call): default bench line, 2-rank rehearsal (gloo, both ranks on the one GPU), rocprofv3 kernel
# stats of the default bench command, PMC HBM-traffic passes of the workloads whose kernels changed, SQ counters of the headline
# in its three forms (rolled stages - the default; two lanes per env; both with ONE launch per step: --no-graph) at 262 144 envs
# and of the mid-size batch where the two-lanes form is AUTO's choice.  Raw output: gpurun_out/r4_p/;
# `python tools/summarize_profile.py r4_p r4_a` turns it into profiles/r4_a/.
set -o pipefail
cd /root/repo
OUT=/root/repo/gpurun_out/r4_p
mkdir -p $OUT
ROBOY_BENCH_BACKEND=gloo timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29655 bench.py --gpus 2 --steps 20 --warmup 5 > $OUT/bench_2rank_gloo.json 2> $OUT/bench_2rank_gloo.err; echo "2-rank rehearsal rc=$?"
timeout -k 10 500 python bench.py > $OUT/bench_unprofiled.json 2> $OUT/bench_unprofiled.err; echo "bench rc=$?"
export TMPDIR=/tmp
cd /tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_stats -- python3 /root/repo/bench.py --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $OUT/prof_stats.err; echo "rocprof stats rc=$?"
pmc() { W=$1; TAG=$2; shift 2
 for C in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 200 rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_${TAG}_$C -- python3 /root/repo/bench.py --no-cpu-baseline --no-also --workload $W --steps 40 --warmup 8 --repeats 1 --no-graph "$@" > /dev/null 2> $OUT/pmc_${TAG}_$C.err; echo "pmc $TAG $C rc=$?"
 done
}
pmc msj-262144-rk4 msj-262144-rk4
pmc msj-2097152-rk4 msj-2097152-rk4
pmc msj-262144-rk4 msj-262144-rk4-pairs --kernel 5
pmc msj-262144-rk4 msj-32768-rk4-pairs --envs 32768
sq() { W=$1; TAG=$2; shift 2; 
  timeout -k 10 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $OUT/pmc_${TAG}_SQ1 -- python3 /root/repo/bench.py --no-cpu-baseline --no-also --workload $W --steps 40 --warmup 8 --repeats 1 --no-graph "$@" > /dev/null 2> $OUT/pmc_${TAG}_SQ1.err; echo "pmc $TAG SQ1 rc=$?"
  timeout -k 10 200 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_${TAG}_SQ2 -- python3 /root/repo/bench.py --no-cpu-baseline --no-also --workload $W --steps 40 --warmup 8 --repeats 1 --no-graph "$@" > /dev/null 2> $OUT/pmc_${TAG}_SQ2.err; echo "pmc $TAG SQ2 rc=$?"
}
sq msj-262144-rk4 msj-262144-rk4
sq msj-262144-rk4 msj-262144-rk4-pairs --kernel 5
sq msj-2097152-rk4 msj-2097152-rk4
sq msj-2097152-rk4 msj-2097152-rk4-pairs --kernel 5
sq msj-262144-rk4 msj-32768-rk4-pairs --envs 32768
sq msj-262144-rk4 msj-32768-rk4-lane --envs 32768 --kernel 1
find $OUT -name "*.csv" | wc -l
