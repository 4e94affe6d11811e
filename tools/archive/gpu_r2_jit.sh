#!/bin/bash
# Round 2: full GPU suite, then the gain of the run-time specialised kernels (MsjRobot with 2 substeps is not the
# ahead-of-time table: JIT on / off)
set -o pipefail
cd /root/repo
OUT=/root/repo/gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -4 $OUT/pytest_gpu.log
[ $rc -eq 0 ] || exit $rc
for J in 1 0; do
 for w in msj-262144-rk4 msj-2097152-euler; do
  ROBOY_SIM_JIT=$J timeout -k 10 200 python bench.py --workload $w --substeps 2 --no-also --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('JIT=$J', '$w', 'substeps 2: launch_us', round(d['roofline']['launch_us_events'],2), 'value', '%.3e'%d['value'])"
 done
done
