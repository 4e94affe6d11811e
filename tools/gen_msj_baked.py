#!/usr/bin/env python3
"""Regenerate gym_roboy_amd/csrc/msj_baked.hpp (MsjRobot's closed-form constants as literals) from the robot
description: builds csrc/gen_msj_baked.cpp with g++ and runs it on MsjRobot.get_description().

    python tools/gen_msj_baked.py [output path]
"""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def generate(path):
    from gym_roboy_amd.envs.robots import MsjRobot
    from build_dir import build_dir          # tools/build_dir.py: outside the repository
    build = build_dir()
    so = os.path.join(build, "libgen_msj_baked.so")
    src = os.path.join(ROOT, "gym_roboy_amd", "csrc", "gen_msj_baked.cpp")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-fPIC", "-shared", "-Wno-unknown-pragmas", "-o", so, src])
    lib = ctypes.CDLL(so)
    desc = MsjRobot().get_description()
    rc = lib.rb_gen_msj_baked(ctypes.byref(desc.as_c_struct()), ctypes.c_double(0.1), 1, path.encode())
    if rc:
        raise RuntimeError("rb_gen_msj_baked failed: %d" % rc)
    return path


if __name__ == "__main__":
    out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gym_roboy_amd", "csrc", "msj_baked.hpp")
    print("wrote", generate(out))
