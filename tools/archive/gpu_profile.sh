#!/bin/bash
# Runs on the GPU box (through gpurun): GPU tests, bench, rocprofv3 kernel stats and PMC passes.
set -o pipefail
cd /root/repo
OUT=/root/repo/gpurun_out
mkdir -p $OUT
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" ; tail -3 $OUT/pytest_gpu.log
timeout -k 10 300 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
export TMPDIR=/tmp
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_stats -- python3 /root/repo/bench.py --no-cpu-baseline > $OUT/prof_stats.json 2> $OUT/prof_stats.err; echo "rocprof stats rc=$?"
for C in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 200 rocprofv3 --pmc $C --output-format csv -d $OUT/prof_pmc_$C -- python3 /root/repo/bench.py --no-cpu-baseline --no-also --workload msj-2097152-euler --steps 40 --warmup 8 --no-graph > $OUT/prof_pmc_$C.json 2> $OUT/prof_pmc_$C.err; echo "pmc $C rc=$?"
done
timeout -k 10 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/prof_pmc_SQ -- python3 /root/repo/bench.py --no-cpu-baseline --no-also --workload msj-2097152-euler --steps 40 --warmup 8 --no-graph > $OUT/prof_pmc_SQ.json 2> $OUT/prof_pmc_SQ.err; echo "pmc SQ rc=$?"
find $OUT -name "*.csv" | head -30
