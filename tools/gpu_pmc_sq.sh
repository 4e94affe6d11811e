#!/bin/bash
# SQ counter passes (separate rocprofv3 --pmc runs, no tracing) over one bench workload's step kernel, eager launches.
#   gpurun -- ./tools/gpu_pmc_sq.sh <tag> <workload> [bench.py args ...]      e.g.  r5_a upper-body-65536-euler --kernel 1
# -> gpurun_out/<tag>/pmc_<workload>_sq{1,2}/ and a printed per-launch mean of every counter
set -o pipefail
TAG=$1; W=$2; shift 2
OUT=/root/repo/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
run() { tag=$1; shift; timeout -k 10 240 rocprofv3 --pmc "$@" --output-format csv -d $OUT/pmc_${W}_$tag -- python3 /root/repo/bench.py --no-cpu-baseline --no-also --workload $W --steps 30 --warmup 5 --repeats 1 --no-graph $EXTRA > /dev/null 2> $OUT/pmc_${W}_$tag.err; echo "pmc $W $tag rc=$?"; }
EXTRA="$*"
run sq1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY &&
run sq2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE || exit 1
python3 - $OUT $W <<'PY'
import csv, glob, collections, json, sys
out, w = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sorted(glob.glob('%s/pmc_%s_sq*' % (out, w))):
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        for row in csv.DictReader(open(f)):
            k = row['Kernel_Name']
            if any(s in k for s in ('tree_lane', 'tree_split', 'msj_step', 'msj_env_step', 'tree_step_aba', 'tree_env_step')):
                acc[k][row['Counter_Name']].append(float(row['Counter_Value']))
res = {}
for k, cs in acc.items():
    res[k] = {c: (sum(v[5:]) / len(v[5:]) if len(v) > 10 else sum(v) / len(v)) for c, v in cs.items()}
    r = res[k]
    print(k[:110])
    for c in sorted(r): print('   %-24s %.1f' % (c, r[c]))
    if 'SQ_WAVE_CYCLES' in r and 'SQ_INSTS_VALU' in r:
        # SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles (MI355X_MICROARCH.md)
        print('   -> cycles per vector instruction per wave: %.2f' % (4 * r['SQ_WAVE_CYCLES'] / r['SQ_INSTS_VALU']))
        for c in ('SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_ACTIVE_INST_VALU', 'SQ_ACTIVE_INST_LDS', 'SQ_ACTIVE_INST_ANY'):
            if c in r: print('   -> %s / SQ_WAVE_CYCLES = %.3f' % (c, r[c] / r['SQ_WAVE_CYCLES']))
json.dump(res, open('%s/sq_%s.json' % (out, w), 'w'), indent=1)
PY
