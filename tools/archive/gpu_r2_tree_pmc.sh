#!/bin/bash
# PMC passes over the joint-tree step kernel (upper body, 8 192 envs, Euler), eager launches
set -o pipefail
OUT=/root/repo/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
W=${W:-upper-body-8192-euler}
run() { tag=$1; shift; timeout -k 10 200 rocprofv3 --pmc "$@" --output-format csv -d $OUT/pmc_tree_$tag -- python3 /root/repo/bench.py --no-cpu-baseline --no-also --workload $W --steps 30 --warmup 5 --repeats 1 --no-graph > /dev/null 2> $OUT/pmc_tree_$tag.err; echo "pmc $tag rc=$?"; }
run sq1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY
run sq2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE
python3 - <<'PY'
import csv,glob,collections
for tag in ('sq1','sq2'):
    for f in glob.glob('/root/repo/gpurun_out/pmc_tree_%s/**/*counter_collection.csv'%tag, recursive=True):
        acc=collections.defaultdict(list)
        for row in csv.DictReader(open(f)):
            if 'tree_step' in row['Kernel_Name']:
                acc[row['Counter_Name']].append(float(row['Counter_Value']))
        for k,v in acc.items():
            v=v[5:] if len(v)>10 else v
            print(tag,k,sum(v)/len(v),len(v))
PY
