#!/usr/bin/env python3
"""Regenerate gym_roboy_amd/csrc/tree_lane_baked.hpp: the env-per-lane acceleration of the committed upper body
(gym_roboy_amd/envs/robots/data/upper_body.json) as straight-line code, written by csrc/tree_lane_gen.hpp
(built with g++ through csrc/gen_tree_lane.cpp).

    python tools/gen_tree_lane_baked.py [output path]
"""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def load_generator():
    from build_dir import build_dir          # tools/build_dir.py: outside the repository
    build = build_dir()
    so = os.path.join(build, "libgen_tree_lane.so")
    csrc = os.path.join(ROOT, "gym_roboy_amd", "csrc")
    deps = [os.path.join(csrc, f) for f in ("gen_tree_lane.cpp", "tree_lane_gen.hpp")]
    if not os.path.exists(so) or any(os.path.getmtime(d) > os.path.getmtime(so) for d in deps):
        subprocess.check_call(["g++", "-O1", "-std=c++17", "-fPIC", "-shared", "-o", so, deps[0]])
    return ctypes.CDLL(so)


def generate(desc, path, lds_c=True, pack=True):
    """Write the generated header of `desc` to `path`; returns (lds_slots, statements, hash, flops of one acceleration, most
    temporaries alive at once).  pack: write structurally identical subtrees (the two arms) as one stream of pair values."""
    lib = load_generator()
    slots, stmts, h, fl, live = ctypes.c_int(0), ctypes.c_int(0), ctypes.c_ulonglong(0), ctypes.c_int(0), ctypes.c_int(0)
    rc = lib.rb_gen_tree_lane(ctypes.byref(desc.as_c_struct()), int(lds_c) | (0 if pack else 2), path.encode(), ctypes.byref(slots),
                              ctypes.byref(stmts), ctypes.byref(h), ctypes.byref(fl), ctypes.byref(live))
    if rc:
        raise RuntimeError("rb_gen_tree_lane failed: %d" % rc)
    return slots.value, stmts.value, h.value, fl.value, live.value


def mates(desc):
    """(mate of every link, mate of every tendon) as the generator pairs them: -1 none, -2 is a mate."""
    lib = load_generator()
    m, tm = (ctypes.c_int * desc.n_q)(), (ctypes.c_int * desc.n_t)()
    rc = lib.rb_gen_tree_lane_mates(ctypes.byref(desc.as_c_struct()), m, tm)
    if rc:
        raise RuntimeError("rb_gen_tree_lane_mates failed: %d" % rc)
    return list(m), list(tm)


SPLIT_HELPERS = 2      # helper waves of the library's ahead-of-time split form (csrc/roboy_sim.hip: RB_SPLIT_HELPERS)
SPLIT_HELPER_SHARE = 80    # percent of a helped part's tendons its helper takes (RB_SPLIT_HELPER_SHARE)
SPLIT_TWO_SWEEPS = 1       # the parts' backward pass in two sweeps around barrier T (RB_SPLIT_TWO_SWEEPS)
SPLIT_SHARE_TRUNK = 1      # one part evaluates the trunk links' inertias / bias forces for all (RB_SPLIT_SHARE_TRUNK)
SPLIT_CUT = 0              # the cut form instead: SPLIT_HELPERS parts as a proximal and a distal wave each (RB_SPLIT_CUT)
SPLIT2_PARTS = 2           # the lean two-part form (RB_SPLIT2_PARTS): no helpers, the trunk shared (RB_SPLIT2_SHARE_TRUNK)
SPLIT2_SHARE_TRUNK = 1


def generate_split(desc, path, max_parts=4, max_helpers=0, helper_share=0, two_sweeps=0, cut=0, share_trunk=0):
    """Write the split-form header (one function per wave) of `desc` to `path`; returns a dict of its figures.
    max_helpers: tendon-helper waves for the longest parts (generate_split in csrc/tree_lane_gen.hpp)."""
    lib = load_generator()
    c = ctypes
    n_parts, part_lds, x_slots, max_stmt, n_stmt, h = c.c_int(0), c.c_int(0), c.c_int(0), c.c_int(0), c.c_int(0), c.c_ulonglong(0)
    n_helpers, helper_stmt = c.c_int(0), c.c_int(0)
    parts = (c.c_int * desc.n_q)()
    rc = lib.rb_gen_tree_lane_split_h(c.byref(desc.as_c_struct()), int(max_parts), int(max_helpers) | (int(helper_share) << 8) | (int(bool(two_sweeps)) << 16) | (int(bool(cut)) << 17) | (int(bool(share_trunk)) << 18), path.encode(), c.byref(n_parts),
                                      c.byref(part_lds), c.byref(x_slots), c.byref(max_stmt), c.byref(n_stmt), parts, c.byref(h),
                                      c.byref(n_helpers), c.byref(helper_stmt))
    if rc:
        raise RuntimeError("rb_gen_tree_lane_split failed: %d" % rc)
    return {"n_parts": n_parts.value, "part_lds": part_lds.value, "x_slots": x_slots.value, "max_stmt": max_stmt.value,
            "n_stmt": n_stmt.value, "part_of_joint": list(parts), "hash": h.value, "n_helpers": n_helpers.value,
            "helper_stmt": helper_stmt.value}


if __name__ == "__main__":
    from gym_roboy_amd.envs.robots import UpperBodyRobot
    out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gym_roboy_amd", "csrc", "tree_lane_baked.hpp")
    print("wrote", out, generate(UpperBodyRobot().get_description(), out))
    out = os.path.join(os.path.dirname(out), "tree_lane_split_baked.hpp")
    print("wrote", out, generate_split(UpperBodyRobot().get_description(), out, max_helpers=SPLIT_HELPERS, helper_share=SPLIT_HELPER_SHARE, two_sweeps=SPLIT_TWO_SWEEPS, cut=SPLIT_CUT, share_trunk=SPLIT_SHARE_TRUNK))
    # the two-part form for the lean layout of tree_lane_split.hpp (two workgroups per CU; csrc/roboy_sim_split2.hip)
    out = os.path.join(os.path.dirname(out), "tree_lane_split2_baked.hpp")
    print("wrote", out, generate_split(UpperBodyRobot().get_description(), out, max_parts=SPLIT2_PARTS, max_helpers=0, share_trunk=SPLIT2_SHARE_TRUNK))
