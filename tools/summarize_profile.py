#!/usr/bin/env python3
"""Turn one tools/gpu_r2_profile.sh run (gpurun_out/<tag>/) into the committed summaries under profiles/<tag>/:
kernel_stats.csv / domain_stats.csv (rocprofv3 --kernel-trace --stats of the default bench command), the bench
JSON lines, hbm_traffic_pmc.json (separate --pmc FETCH_SIZE / WRITE_SIZE passes per workload, FETCH_SIZE
doubled as MI355X_MICROARCH.md's HBM section prescribes for gfx950) and sq_counters.json.

    python tools/summarize_profile.py r3_a
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) < 2:
    sys.exit("usage: summarize_profile.py <raw run tag under gpurun_out/> [<directory under profiles/>]   (no default: an earlier round's "
             "committed summaries are not to be overwritten by accident)")
tag = sys.argv[1]
SRC = os.path.join(ROOT, "gpurun_out", tag)
DST = os.path.join(ROOT, "profiles", sys.argv[2] if len(sys.argv) > 2 else tag)       # (raw run directory, committed directory)
os.makedirs(DST, exist_ok=True)
STEP_KERNELS = ("msj_step_env_per_lane", "msj_step_tendon_per_lane", "msj_step_mirror_pairs", "msj_env_step_mirror_pairs", "msj_env_step_tendon_per_lane", "tree_step_aba", "msj_env_step_kernel",
                "tree_lane_step", "tree_lane_env_step", "tree_env_step_aba", "tree_split_step", "tree_split_env_step")


# the kernel sources the counters were collected on (gpu_profile_round.sh writes the hash on the GPU box): every row carries it, and
# bench.py marks a row whose stamp differs from the tree's hash as stale
try:
    with open(os.path.join(SRC, "csrc_hash.txt")) as fh:
        CSRC_HASH = fh.read().split()[0]
except Exception:
    CSRC_HASH = None
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import bench  # noqa: E402
import code_object_meta  # noqa: E402
if CSRC_HASH != bench.csrc_hash():
    print("WARNING: the raw run was collected on kernel sources %s, this tree has %s: rows will read as stale" % (CSRC_HASH, bench.csrc_hash()))
try:
    # registers / scratch / static LDS as the CODE OBJECTS state them (rocprofv3's own trace columns are the dispatch packet's granulated
    # values: 'VGPR 236' for a kernel whose metadata says 470); only meaningful when the library in the tree is the one that was profiled
    CO_META = {code_object_meta.short(k): v for k, v in code_object_meta.kernel_metadata().items()}
except Exception as exc:
    print("no code-object metadata (%s): resource columns left empty" % exc)
    CO_META = {}


# Dynamic LDS is a launch argument (neither in the code object nor in rocprofv3's trace columns, which read 0 for it): what the library's
# launches pass for the committed upper body's ahead-of-time kernels (tree_lane.hpp: LDS_BYTES_PER_WAVE = (107 + 2 * 20) slots * 256 B;
# tree_lane_split.hpp: SP_LDS_BYTES of the five-wave form = 145.5 KB, of the lean two-part form = 55.5 KB; the host's formulas are
# static_assert'ed against the kernels' in csrc/)
DYNAMIC_LDS = {"rbl_baked::tree_lane_step": 37632, "rbl_baked::tree_lane_env_step": 37632,
               "rbl_split_baked::tree_split_step": 148992, "rbl_split_baked::tree_split_env_step": 148992,
               "rbl_split2_baked::tree_split_step": 56832, "rbl_split2_baked::tree_split_env_step": 56832}


def counters(dirname):
    """{kernel name: {counter: [values per dispatch]}} of one pass, step kernels only"""
    out = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(SRC, dirname, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if any(k in row["Kernel_Name"] for k in STEP_KERNELS):
                out[row["Kernel_Name"]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    return out


def mean_tail(v, skip=10):
    v = v[skip:] if len(v) > 2 * skip else v
    return sum(v) / len(v), len(v)


for name in ("bench_unprofiled.json", "bench_under_rocprof.json", "bench_2rank_gloo.json", "bench_also_unprofiled.json", "train_rollout_world1.json",
             "train_rollout_world2.json", "ppo_update_kernel_stats.txt"):
    p = os.path.join(SRC, name)
    if os.path.exists(p):
        shutil.copy(p, os.path.join(DST, name))
for f in glob.glob(os.path.join(SRC, "prof_stats", "**", "*_stats.csv"), recursive=True):
    base = os.path.basename(f).split("_", 1)[1]
    shutil.copy(f, os.path.join(DST, base))

# The step kernels per (symbol, grid size): one symbol serves several workloads of the bench run (the RK4 env-per-lane kernel
# runs the 262 144-env headline and the 2 097 152-env batch), so rocprofv3's per-symbol average mixes them; the per-dispatch
# trace has the grid size, which tells them apart.  Also the registers / scratch / LDS the dispatches were launched with.
by_grid = collections.defaultdict(list)
meta = {}
for f in glob.glob(os.path.join(SRC, "prof_stats", "**", "*kernel_trace.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        name = row["Kernel_Name"]
        if not any(k in name for k in STEP_KERNELS):
            continue
        key = (name, int(row["Grid_Size_X"]), int(row["Workgroup_Size_X"]))
        by_grid[key].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
        meta[key] = (row["LDS_Block_Size"],)            # the dispatch's LDS (static + dynamic): a launch argument, not in the code object
if by_grid:
    with open(os.path.join(DST, "kernel_stats_by_grid.csv"), "w") as fh:
        w = csv.writer(fh)
        w.writerow(["Name", "Grid_Size_X", "Workgroup_Size_X", "Envs", "Calls", "AverageNs", "MinNs", "MaxNs", "VGPR", "AGPR", "SGPR", "ScratchBytesPerLane",
                    "VGPRSpills", "SGPRSpills", "StaticLDSBytes", "DynamicLDSBytesOfTheLaunch", "csrc_hash"])
        for key in sorted(by_grid, key=lambda k: (k[0], k[1])):
            v = by_grid[key]
            tail = v[10:] if len(v) > 20 else v
            # threads per env: 8 in the tendon-per-lane form, 32 in the octet kernels (2 envs per wave), 1 otherwise
            per_env = 8 if "tendon_per_lane" in key[0] else 2 if "mirror_pairs" in key[0] else (32 if "_aba" in key[0] else (key[2] // 64 if "tree_split" in key[0] else 1))
            co = CO_META.get(code_object_meta.short(key[0])) or {}
            w.writerow([key[0], key[1], key[2], key[1] // per_env, len(v), "%.1f" % (sum(tail) / len(tail)), min(tail), max(tail),
                        co.get("vgpr_count", ""), co.get("agpr_count", ""), co.get("sgpr_count", ""), co.get("private_segment_fixed_size", ""),
                        co.get("vgpr_spill_count", ""), co.get("sgpr_spill_count", ""), co.get("group_segment_fixed_size", ""),
                        DYNAMIC_LDS.get(code_object_meta.short(key[0]).split("<")[0], 0), CSRC_HASH])

traffic = {}
for d in sorted(glob.glob(os.path.join(SRC, "pmc_*_FETCH_SIZE"))):
    w = os.path.basename(d)[len("pmc_"):-len("_FETCH_SIZE")]
    fetch = counters(os.path.basename(d))
    write = counters("pmc_%s_WRITE_SIZE" % w)
    if not fetch or not write:
        continue
    # the dominant step kernel of the pass = the one with the most dispatches
    kern = max(fetch, key=lambda k: len(fetch[k]["FETCH_SIZE"]))
    fm, fn = mean_tail(fetch[kern]["FETCH_SIZE"])
    wm, wn = mean_tail(write[kern]["WRITE_SIZE"])
    traffic[w] = {"FETCH_SIZE": {"dispatches": fn, "mean_KiB": fm}, "WRITE_SIZE": {"dispatches": wn, "mean_KiB": wm},
                  "kernel": kern, "hbm_bytes_per_launch": (2.0 * fm + wm) * 1024.0, "csrc_hash": CSRC_HASH,
                  "correction": "FETCH_SIZE x2 (gfx950 tallies 128-B read requests at 64 B, MI355X_MICROARCH.md HBM section); "
                                "WRITE_SIZE as reported; KiB -> bytes"}
with open(os.path.join(DST, "hbm_traffic_pmc.json"), "w") as fh:
    json.dump(traffic, fh, indent=1, sort_keys=True)
    fh.write("\n")

sq = {}
for d in sorted(glob.glob(os.path.join(SRC, "pmc_*_SQ1"))):
    w = os.path.basename(d)[len("pmc_"):-len("_SQ1")]
    merged = collections.defaultdict(dict)
    for part in ("SQ1", "SQ2"):
        for kern, cs in counters("pmc_%s_%s" % (w, part)).items():
            for cname, vals in cs.items():
                merged[kern][cname] = mean_tail(vals)[0]
    if not merged:
        continue
    kern = max(merged, key=lambda k: merged[k].get("SQ_WAVES", 0))
    c = merged[kern]
    waves = c.get("SQ_WAVES", 0) or 1.0
    per_wave = {k: c[k] / waves for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_WAVE_CYCLES") if k in c}
    frac = {k: c[k] / c["SQ_WAVE_CYCLES"] for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU")
            if k in c and c.get("SQ_WAVE_CYCLES")}
    if c.get("SQ_LDS_IDX_ACTIVE"):
        frac["lds_bank_conflict_of_lds_active"] = c.get("SQ_LDS_BANK_CONFLICT", 0.0) / c["SQ_LDS_IDX_ACTIVE"]
    sq[w] = {"kernel": kern, "csrc_hash": CSRC_HASH, "per_launch": dict(c), "per_wave": per_wave, "fraction_of_wave_cycles": frac,
             "note": "SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles (MI355X_MICROARCH.md, cycle constants)"}
with open(os.path.join(DST, "sq_counters.json"), "w") as fh:
    json.dump(sq, fh, indent=1, sort_keys=True)
    fh.write("\n")
print("wrote", DST, sorted(os.listdir(DST)))
for w, t in traffic.items():
    print("%-24s %10.0f B/launch  %s" % (w, t["hbm_bytes_per_launch"], t["kernel"][:70]))
for w, s in sq.items():
    print(w, {k: round(v, 1) for k, v in s["per_wave"].items()}, {k: round(v, 3) for k, v in s["fraction_of_wave_cycles"].items()})
