#!/bin/bash
# PMC passes over the env-per-lane joint-tree step kernel (upper body, Euler), eager launches: where does a wave's time go?
set -o pipefail
OUT=/root/repo/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 -L > $OUT/counters_list.txt 2>&1
grep -o "SQC_[A-Z_0-9]*\|SQ_IFETCH[A-Z_0-9]*\|SQ_INST_LEVEL[A-Z_0-9]*\|SQ_WAIT_IFETCH[A-Z_0-9]*" $OUT/counters_list.txt | sort -u | tr '\n' ' '; echo
W=${W:-upper-body-8192-euler}
run() { n=$1; tag=$2; shift 2; timeout -k 10 200 rocprofv3 --pmc "$@" --output-format csv -d $OUT/pmc_lane_${n}_$tag -- python3 /root/repo/bench.py --no-cpu-baseline --no-also --workload $W --envs $n --kernel 1 --steps 30 --warmup 5 --repeats 1 --no-graph > /dev/null 2> $OUT/pmc_lane_${n}_$tag.err; echo "pmc $n $tag rc=$?"; }
for n in 64 65536; do
run $n sq1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY &&
run $n sq2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM SQ_ACTIVE_INST_ANY SQ_IFETCH SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE &&
run $n sq3 SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_INSTS_SMEM SQ_INSTS_FLAT SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM || exit 1
done
python3 - <<'PY'
import csv,glob,collections
for d in sorted(glob.glob('/root/repo/gpurun_out/pmc_lane_*')):
    if not d.endswith(('sq1','sq2','sq3')): continue
    for f in glob.glob(d+'/**/*counter_collection.csv', recursive=True):
        acc=collections.defaultdict(list)
        for row in csv.DictReader(open(f)):
            if 'tree_lane' in row['Kernel_Name']:
                acc[row['Counter_Name']].append(float(row['Counter_Value']))
        for k,v in acc.items():
            v=v[5:] if len(v)>10 else v
            print(d.split('pmc_lane_')[1],k,sum(v)/len(v),len(v))
PY
