#!/usr/bin/env python3
"""Instruction histogram of the library's step kernels from the compiler's own assembly (hipcc -save-temps of
csrc/roboy_sim.hip, no GPU needed): per kernel the counts by class - v_fma / v_mul / v_add / v_mov / v_cndmask /
transcendental / other VALU / AGPR moves / SALU / LDS / VMEM / waits - plus VGPR, AGPR, scratch and LDS sizes from the
kernel descriptor comments.  Writes JSON (default: stdout).

    python tools/isa_histogram.py [--out profiles/r3_a/isa_histogram.json] [--match REGEX ...]
"""
import argparse
import collections
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "gym_roboy_amd", "csrc")
DEFAULT = [r"msj_step_env_per_lane.*Li1ELi256ELi4ELb1E", r"msj_step_env_per_lane.*Li0ELi256ELi4ELb1E",
           r"tree_lane_stepILi0E", r"tree_lane_stepILi1E", r"tree_lane_env_stepILi0E", r"tree_split_stepILi0E", r"tree_split_stepILi1E",
           r"tree_step_abaILi0ELi2ELb1E"]
TRANS = ("v_exp_f32", "v_log_f32", "v_rcp_f32", "v_rsq_f32", "v_sqrt_f32", "v_sin_f32", "v_cos_f32")


def classify(op):
    if op.startswith(("v_fma", "v_fmac", "v_mad_f32", "v_fmamk", "v_fmaak", "v_pk_fma")):
        return "v_fma"
    if op.startswith(("v_mul_f32", "v_pk_mul_f32")):
        return "v_mul"
    if op.startswith(("v_add_f32", "v_sub_f32", "v_subrev_f32", "v_pk_add_f32")):
        return "v_add"
    if op.startswith("v_accvgpr"):
        return "agpr_move"
    if op.startswith(("v_mov", "v_pk_mov")):
        return "v_mov"
    if op.startswith("v_cndmask"):
        return "v_cndmask"
    if op.startswith(TRANS):
        return "transcendental"
    if op.startswith(("v_min", "v_max", "v_med3")):
        return "v_minmax"
    if op.startswith(("v_readlane", "v_writelane", "v_readfirstlane")):
        return "lane_move"
    if op.startswith("v_"):
        return "v_other"
    if op.startswith("s_waitcnt"):
        return "s_waitcnt"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "scratch" if op.startswith("scratch_") else "vmem"
    return "other"


def histogram(asm, patterns):
    out = {}
    blocks = re.split(r"\n(?=[A-Za-z_][A-Za-z_0-9$.]*:\s*(?:;.*)?\n)", asm)
    for blk in blocks:
        name = blk.split(":", 1)[0].strip()
        if not name.startswith("_Z") or "s_endpgm" not in blk:
            continue
        if not any(re.search(p, name) for p in patterns):
            continue
        body = blk.split("s_endpgm")[0]
        ops = re.findall(r"^\s+([a-z][a-z_0-9]+)", body, flags=re.M)
        cls = collections.Counter(classify(o) for o in ops)
        detail = collections.Counter(ops)
        meta = {}
        tail = blk.split("s_endpgm", 1)[1]
        for key, pat in (("vgprs", r"; NumVgprs: (\d+)"), ("agprs", r"; NumAgprs: (\d+)"), ("sgprs", r"; NumSgprs: (\d+)"),
                         ("scratch_bytes", r"; ScratchSize: (\d+)"), ("occupancy_waves_per_simd", r"; Occupancy: (\d+)"),
                         ("code_bytes", r"; codeLenInByte = (\d+)")):
            m = re.search(pat, tail)
            if m:
                meta[key] = int(m.group(1))
        valu = sum(v for k, v in cls.items() if k.startswith("v_") or k in ("transcendental", "agpr_move", "lane_move"))
        out[name] = {"classes": dict(sorted(cls.items())), "valu_total": valu, "instructions": len(ops), "resources": meta,
                     "top": dict(detail.most_common(25))}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=None)
    ap.add_argument("--match", nargs="*", default=DEFAULT)
    args = ap.parse_args()
    with tempfile.TemporaryDirectory() as tmp:
        cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-fno-slp-vectorize", "-S",
               "--cuda-device-only", "-o", os.path.join(tmp, "roboy_sim.s"), os.path.join(CSRC, "roboy_sim.hip")]
        subprocess.check_call(cmd, stderr=subprocess.DEVNULL)
        asm = open(os.path.join(tmp, "roboy_sim.s")).read()
    h = histogram(asm, args.match)
    text = json.dumps({"_about": "tools/isa_histogram.py: static instruction counts of the compiled step kernels "
                                 "(hipcc -O3 --offload-arch=gfx950 -fno-slp-vectorize, the library's flags); loops count once",
                       "kernels": h}, indent=1, sort_keys=True)
    if args.out:
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        open(args.out, "w").write(text + "\n")
        for k, v in h.items():
            print(k[:70], v["valu_total"], v["classes"], v["resources"])
    else:
        print(text)


if __name__ == "__main__":
    main()
