#!/bin/bash
# Round 2: full GPU suite, then HBM-traffic passes of the joint-tree workloads (after a layout change)
set -o pipefail
cd /root/repo
OUT=/root/repo/gpurun_out/r2_a
mkdir -p $OUT
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/../pytest_gpu.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $OUT/../pytest_gpu.log
[ $rc -eq 0 ] || exit $rc
export TMPDIR=/tmp
cd /tmp
for W in upper-body-8192-euler upper-body-8192-rk4; do
 for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf $OUT/pmc_${W}_$C
  timeout -k 10 200 rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_${W}_$C -- python3 /root/repo/bench.py --no-cpu-baseline --no-also --workload $W --steps 40 --warmup 8 --repeats 1 --no-graph > /dev/null 2> $OUT/pmc_${W}_$C.err; echo "pmc $W $C rc=$?"
 done
done
W=upper-body-8192-euler
rm -rf $OUT/pmc_${W}_SQ1 $OUT/pmc_${W}_SQ2
timeout -k 10 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $OUT/pmc_${W}_SQ1 -- python3 /root/repo/bench.py --no-cpu-baseline --no-also --workload $W --steps 40 --warmup 8 --repeats 1 --no-graph > /dev/null 2> $OUT/pmc_${W}_SQ1.err; echo "pmc $W SQ1 rc=$?"
timeout -k 10 200 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_${W}_SQ2 -- python3 /root/repo/bench.py --no-cpu-baseline --no-also --workload $W --steps 40 --warmup 8 --repeats 1 --no-graph > /dev/null 2> $OUT/pmc_${W}_SQ2.err; echo "pmc $W SQ2 rc=$?"
cd /root/repo
for w in upper-body-8192-euler upper-body-8192-rk4; do
  timeout -k 10 200 python bench.py --workload $w --no-also --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$w', 'launch_us', round(d['roofline']['launch_us_events'],2), 'value', '%.3e'%d['value'])"
done
