#!/bin/bash
# round 4: instruction-cache counters of the split joint-tree kernels (every wave of a workgroup runs a function of its own:
# the code one acceleration walks through is ~80 KB against 64 KB of instruction cache per CU pair)
set -o pipefail
cd /root/repo
OUT=/root/repo/gpurun_out/r4_p
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for W in upper-body-8192-euler upper-body-8192-rk4; do
  timeout -k 10 200 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY --output-format csv -d $OUT/pmc_${W}_IC -- python3 /root/repo/bench.py --no-cpu-baseline --no-also --workload $W --steps 40 --warmup 8 --repeats 1 --no-graph > /dev/null 2> $OUT/pmc_${W}_IC.err; echo "pmc $W IC rc=$?"
  timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_IFETCH_LEVEL SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_${W}_IC2 -- python3 /root/repo/bench.py --no-cpu-baseline --no-also --workload $W --steps 40 --warmup 8 --repeats 1 --no-graph > /dev/null 2> $OUT/pmc_${W}_IC2.err; echo "pmc $W IC2 rc=$?"
done
python3 - <<'PY'
import csv, glob, collections
for d in sorted(glob.glob('/root/repo/gpurun_out/r4_p/pmc_upper-body-8192-*_IC*')):
    if d.endswith('.err'): continue
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name'][:60]
            agg[k][r['Counter_Name']] += float(r['Counter_Value']); 
            if r['Counter_Name'] in ('SQ_WAVE_CYCLES', 'SQ_INSTS_VALU'): n[k] += 1
    for k, v in agg.items():
        if 'split' in k: print(d.split('/')[-1], k, n[k], {c: round(x / max(n[k], 1)) for c, x in v.items()})
PY
