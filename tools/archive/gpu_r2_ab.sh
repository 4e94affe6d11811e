#!/bin/bash
# Round 2: parity of the rebuilt library, then the A/B of the large-batch launch configurations.
set -o pipefail
cd /root/repo
OUT=/root/repo/gpurun_out
mkdir -p $OUT
timeout -k 10 900 python -m pytest tests -m gpu -x -q --deselect tests/test_bench_contract_gpu.py > $OUT/pytest_gpu.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $OUT/pytest_gpu.log
[ $rc -eq 0 ] || exit $rc
WL="${WL:-msj-262144-rk4 msj-2097152-euler}" EXTRA="${EXTRA:-}" timeout -k 10 900 bash tools/ab.sh > $OUT/ab.log 2>&1
cat $OUT/ab.log
