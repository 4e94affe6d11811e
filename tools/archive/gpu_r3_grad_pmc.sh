#!/bin/bash
# PMC passes over the PPO gradient kernels (tools/policy_grad_profile.py: 8 388 608-sample minibatch, indexed then contiguous)
set -o pipefail
OUT=/root/repo/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 -L > $OUT/counters_list.txt 2>&1
grep -o "SQ_[A-Z_0-9]*MFMA[A-Z_0-9]*\|SQ_LDS[A-Z_0-9]*\|SQ_INSTS_[A-Z_0-9]*" $OUT/counters_list.txt | sort -u | tr '\n' ' '; echo
run() { tag=$1; shift; timeout -k 10 300 rocprofv3 --pmc "$@" --output-format csv -d $OUT/pmc_grad_$tag -- python3 /root/repo/tools/policy_grad_profile.py > /dev/null 2> $OUT/pmc_grad_$tag.err; echo "pmc $tag rc=$?"; }
run sq1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY &&
run sq2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE &&
run sq3 SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_ACTIVE_INST_MISC SQ_INSTS_FLAT SQ_INSTS_SMEM
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/grad_stats -- python3 /root/repo/tools/policy_grad_profile.py > /dev/null 2> $OUT/grad_stats.err; echo "stats rc=$?"
python3 - <<'PY'
import csv,glob,collections
for d in sorted(glob.glob('/root/repo/gpurun_out/pmc_grad_*')):
    if not d.endswith(('sq1','sq2','sq3')): continue
    for f in glob.glob(d+'/**/*counter_collection.csv', recursive=True):
        acc=collections.defaultdict(list)
        for row in csv.DictReader(open(f)):
            if 'mlp_grad' in row['Kernel_Name']:
                acc[(row['Kernel_Name'][33:60], row['Counter_Name'])].append(float(row['Counter_Value']))
        for k,v in sorted(acc.items()):
            print(d.split('pmc_grad_')[1],k,' '.join('%.4g'%x for x in v))
for f in glob.glob('/root/repo/gpurun_out/grad_stats/**/*kernel_stats.csv', recursive=True):
    for row in csv.DictReader(open(f)):
        if 'mlp_grad' in row['Name'] or 'reduce' in row['Name']: print(row['Name'][:70], row['Calls'], row['AverageNs'], row['MinNs'], row['MaxNs'])
PY
