#!/bin/bash
# Round 4 profile pass, second part: the split joint-tree kernels with tendon-helper waves (configs[3]) - PMC traffic and SQ counters;
# refreshed kernel stats of the default bench command.  Raw output joins gpurun_out/r4_p/.
set -o pipefail
cd /root/repo
OUT=/root/repo/gpurun_out/r4_p
mkdir -p $OUT
timeout -k 10 500 python bench.py > $OUT/bench_unprofiled.json 2> $OUT/bench_unprofiled.err; echo "bench rc=$?"
export TMPDIR=/tmp
cd /tmp
rm -rf $OUT/prof_stats
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_stats -- python3 /root/repo/bench.py --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $OUT/prof_stats.err; echo "rocprof stats rc=$?"
pmc() { W=$1; TAG=$2; shift 2
 for C in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 200 rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_${TAG}_$C -- python3 /root/repo/bench.py --no-cpu-baseline --no-also --workload $W --steps 40 --warmup 8 --repeats 1 --no-graph "$@" > /dev/null 2> $OUT/pmc_${TAG}_$C.err; echo "pmc $TAG $C rc=$?"
 done
}
pmc upper-body-8192-euler upper-body-8192-euler
pmc upper-body-8192-rk4 upper-body-8192-rk4
for C in FETCH_SIZE WRITE_SIZE; do
  VECENV_ROBOT=upper VECENV_SIZES=8192 timeout -k 10 200 rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_fused-env-UpperBodyRobot-8192_$C -- python3 /root/repo/tools/vecenv_bench.py > /dev/null 2> $OUT/pmc_fused-env-upper8k_$C.err; echo "pmc fused-env upper 8192 $C rc=$?"
done
sq() { W=$1; TAG=$2; shift 2; 
  timeout -k 10 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $OUT/pmc_${TAG}_SQ1 -- python3 /root/repo/bench.py --no-cpu-baseline --no-also --workload $W --steps 40 --warmup 8 --repeats 1 --no-graph "$@" > /dev/null 2> $OUT/pmc_${TAG}_SQ1.err; echo "pmc $TAG SQ1 rc=$?"
  timeout -k 10 200 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_${TAG}_SQ2 -- python3 /root/repo/bench.py --no-cpu-baseline --no-also --workload $W --steps 40 --warmup 8 --repeats 1 --no-graph "$@" > /dev/null 2> $OUT/pmc_${TAG}_SQ2.err; echo "pmc $TAG SQ2 rc=$?"
}
sq upper-body-8192-euler upper-body-8192-euler
sq upper-body-8192-rk4 upper-body-8192-rk4
find $OUT -name "*.csv" | wc -l
