"""CPU checks of the drop-in boundary: the C-ABI library loads, exports every
symbol include/roboy_sim.h declares, mirrors the structs, and fails loudly
(error code + message, no abort, no CPU fallback) when no GPU is present."""
import ctypes
import os
import re

import numpy as np
import pytest

from gym_roboy_amd import _native as nat
from gym_roboy_amd.envs.robots import MsjRobot, RobotDescription, msj_platform_spec

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = open(os.path.join(ROOT, "include", "roboy_sim.h")).read()


def test_library_exports_every_declared_symbol():
    declared = set(re.findall(r"\b(rb_[a-z0-9_]+)\s*\(", HEADER))
    assert len(declared) >= 25
    assert declared == set(nat.SIGNATURES), declared ^ set(nat.SIGNATURES)
    lib = nat.load()
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.rb_abi_version() == int(re.search(r"#define RB_ABI_VERSION (\d+)", HEADER).group(1))


def test_struct_mirrors_have_the_c_layout():
    from gym_roboy_amd.envs.robots.description import RobotDescriptionC
    # rb_robot_desc: 4 int32 + 11 pointers + 3 doubles + 4 pointers + 8 doubles
    assert ctypes.sizeof(RobotDescriptionC) == 16 + 11 * 8 + 24 + 4 * 8 + 8 * 8
    assert ctypes.sizeof(nat.SimInfo) == 8 + 6 * 4 + 8 + 8 + 8
    assert ctypes.sizeof(nat.EnvConfig) == 4 * 4 + 10 * 4
    header_fields = re.search(r"typedef struct rb_env_config \{(.*?)\} rb_env_config;", HEADER, re.S).group(1)
    names = re.findall(r"\b(?:int32_t|float)\s+([a-z_, ]+);", header_fields)
    flat = [n.strip() for group in names for n in group.split(",")]
    assert flat == [f for f, _ in nat.EnvConfig._fields_]


def test_description_roundtrip_and_validation(tmp_path):
    d = MsjRobot.get_description()
    assert (d.n_q, d.n_t, d.n_vp) == (3, 8, 24)
    p = tmp_path / "msj.json"
    d.to_json(str(p))
    d2 = RobotDescription.from_json(str(p))
    assert np.array_equal(d.vp_pos, d2.vp_pos) and d.muscle == d2.muscle
    c = d.as_c_struct()
    assert c.n_q == 3 and c.vp_offset[8] == 24 and abs(c.gravity[2] + 9.81) < 1e-12
    committed = RobotDescription.from_json(os.path.join(ROOT, "gym_roboy_amd", "envs", "robots", "data", "msj_platform.json"))
    assert np.array_equal(committed.vp_pos, d.vp_pos) and np.array_equal(committed.f_max, d.f_max)
    bad = msj_platform_spec(); bad["format"] = "something/else"
    with pytest.raises(ValueError):
        RobotDescription(bad)
    bad = msj_platform_spec(); bad["joints"][1]["parent"] = 2
    with pytest.raises(ValueError):
        RobotDescription(bad)
    bad = msj_platform_spec(); bad["tendons"][0]["via_points"] = bad["tendons"][0]["via_points"][:1]
    with pytest.raises(ValueError):
        RobotDescription(bad)
    bad = msj_platform_spec(); bad["muscle"]["typo"] = 1.0
    with pytest.raises(ValueError):
        RobotDescription(bad)


def test_argument_errors_are_reported_not_fatal():
    lib = nat.load()
    d = MsjRobot.get_description()
    h = ctypes.c_void_p()
    rc = lib.rb_create(ctypes.byref(d.as_c_struct()), 0, 0, 0.1, 1, 0, 0, 0, ctypes.byref(h))
    assert rc == nat.RB_EINVAL and b"n_envs" in lib.rb_last_error()
    rc = lib.rb_create(ctypes.byref(d.as_c_struct()), 4, 7, 0.1, 1, 0, 0, 0, ctypes.byref(h))
    assert rc == nat.RB_EINVAL and b"integrator" in lib.rb_last_error()
    assert lib.rb_synchronize(None) == nat.RB_EINVAL
    with pytest.raises(ValueError):
        nat.check(lib.rb_step_dev(None, None, 1.0))


@pytest.mark.skipif(nat.device_count() > 0, reason="a GPU is present")
def test_without_a_gpu_the_product_path_fails_loudly():
    from gym_roboy_amd.envs.simulations import HipBatchSimulation, HipSimulationClient
    with pytest.raises(nat.NativeError, match="no HIP device"):
        HipBatchSimulation(MsjRobot(), 16)
    with pytest.raises(nat.NativeError):
        HipSimulationClient(MsjRobot())


def test_product_never_imports_the_oracle():
    """The oracle is test infrastructure: nothing under gym_roboy_amd/ may name it
    (the oracle-backed CpuSimulationClient lives in oracle/ for that reason)."""
    pkg = os.path.join(ROOT, "gym_roboy_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                for line in text.splitlines():
                    s = line.strip()
                    if s.startswith(("import ", "from ", "#include")):
                        assert "oracle" not in s, (f, s)


def test_bench_never_prints_outside_its_json_line():
    """bench.py's stdout is one JSON line; anything that can print (the RoboyEnv
    goal banner in the python-loop baseline) must run under redirect_stdout."""
    src = open(os.path.join(ROOT, "bench.py")).read()
    # the Python env workers are child processes whose stdout is a pipe; inside them the env runs
    # under redirect_stdout and only the "<steps> <seconds>" result line is written
    body = src[src.index("_PY_ENV_WORKER"):src.index("def cpu_python_env_loop")]
    guarded = body[body.index("with contextlib.redirect_stdout"):]
    assert "RoboyEnv(simulation_client" in guarded and "RoboyEnv(simulation_client" not in body[:body.index("with contextlib.redirect_stdout")]
    assert "stdout=subprocess.PIPE" in src[src.index("def cpu_python_env_loop"):src.index("def cpu_baseline")]
    # the only writer to the real stdout is the os.write of the JSON line on the saved descriptor;
    # fd 1 itself points at stderr for the whole run (RCCL prints a banner on stdout)
    assert src.count("print(") == 0
    assert src.count("os.write(json_fd") == 1 and "os.dup2(2, 1)" in src


def test_committed_baked_constant_table_is_current(tmp_path):
    """gym_roboy_amd/csrc/msj_baked.hpp (MsjRobot's closed-form constants as literals, used by the BK kernel
    instances) must be what csrc/gen_msj_baked.cpp produces from the robot description today."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import gen_msj_baked
    fresh = gen_msj_baked.generate(str(tmp_path / "msj_baked.hpp"))
    committed = os.path.join(ROOT, "gym_roboy_amd", "csrc", "msj_baked.hpp")
    assert open(fresh).read() == open(committed).read(), "run python tools/gen_msj_baked.py and rebuild"
