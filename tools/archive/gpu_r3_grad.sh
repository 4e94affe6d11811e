#!/bin/bash
# gradient kernels: policy tests, then kernel times with and without the LDS-DMA prefetch (tools/policy_grad_profile.py)
set -o pipefail
OUT=/root/repo/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd /root/repo
timeout -k 10 600 python -m pytest tests/test_policy_gpu.py tests/test_ppo.py -x -q -m gpu 2>&1 | tail -5 || exit 1
cd /tmp
for pf in 1 0; do
ROBOY_POLICY_PREFETCH=$pf timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/grad_stats_pf$pf -- python3 /root/repo/tools/policy_grad_profile.py > /dev/null 2> $OUT/grad_stats_pf$pf.err; echo "stats pf=$pf rc=$?"
python3 - $pf <<'PY'
import csv,glob,sys
for f in glob.glob('/root/repo/gpurun_out/grad_stats_pf%s/**/*kernel_trace.csv' % sys.argv[1], recursive=True):
    rows = [r for r in csv.DictReader(open(f)) if 'mlp_grad' in r['Kernel_Name']]
    for r in rows: print(r['Kernel_Name'][30:62], (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3, 'us')
PY
done
