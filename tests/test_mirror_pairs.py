"""CPU tests of the "mirror pairs" form (csrc/msj_kernels.hpp: two lanes per env): the mirror-plane detection
(csrc/msj_build.hpp: find_mirror_pairs) and the arithmetic - the even lane's half of the tendons from the env's state,
the odd lane's half from the MIRROR IMAGE of that state with the same constants, the two torque sums combined as
pseudovectors, the rolled RK4 of integrate_acc - emulated on the host (tests/hostmath) against the fp64 oracle.
The DPP swap between the lanes is the only piece the GPU tests (test_physics_gpu.py) add."""
import copy
import ctypes

import numpy as np
import pytest

from conftest import random_states
from oracle.physics_np import TendonRobotOracle


def _pairs(lib, desc, q, qd, sp, integ, nsub=1, dtype=np.float64):
    P = lambda a, t: a.ctypes.data_as(ctypes.POINTER(t))
    ct = ctypes.c_double if dtype == np.float64 else ctypes.c_float
    q1, qd1, sp1 = q.astype(dtype), qd.astype(dtype), sp.astype(dtype)
    f1 = np.zeros(len(q), np.uint8)
    mirror = ctypes.c_int(-1)
    half = (ctypes.c_int * 8)()
    fn = lib.hm_step_pairs_f64 if dtype == np.float64 else lib.hm_step_pairs_f32
    rc = fn(ctypes.byref(desc.as_c_struct()), ctypes.c_double(0.1), nsub, integ, ctypes.c_long(len(q)),
            P(q1, ct), P(qd1, ct), P(sp1, ct), P(f1, ctypes.c_ubyte), ctypes.byref(mirror), half)
    return rc, q1, qd1, f1.astype(bool), mirror.value, list(half)


def _rotated_msj():
    """MsjRobot turned by 90 degrees about z ((x, y) -> (-y, x), exact in floating point): its mirror plane is then the
    y-z plane, and only that one (the strong / weak tendon pairs sit on the +y / -y side)."""
    from gym_roboy_amd.envs.robots import RobotDescription, msj_platform_spec
    spec = copy.deepcopy(msj_platform_spec())
    for t in spec["tendons"]:
        for v in t["via_points"]:
            x, y, z = v["pos"]
            v["pos"] = [-y, x, z]
    return RobotDescription(spec)


@pytest.mark.parametrize("integ", [0, 1])
@pytest.mark.parametrize("nsub", [1, 3])
def test_msj_robot_stepped_as_mirror_pairs_matches_the_oracle(msj_robot, msj_oracle, hostmath_lib, integ, nsub):
    desc = msj_robot.get_description()
    q, qd, sp = random_states(desc, 1200, 31)
    want = msj_oracle.step(q.astype(np.float64), qd.astype(np.float64), sp.astype(np.float64), integrator=integ, n_substeps=nsub)
    rc, q1, qd1, f1, mirror, half = _pairs(hostmath_lib, desc, q, qd, sp, integ, nsub)
    assert rc == 0 and mirror == 0                         # the x-z plane: tendon k <-> tendon 7 - k
    assert half == [0, 1, 2, 3, 7, 6, 5, 4]
    assert np.abs(q1 - want[0]).max() < 1e-12 and np.abs(qd1 - want[1]).max() < 1e-11
    assert np.array_equal(f1, want[2])
    rc, q2, qd2, f2, _, _ = _pairs(hostmath_lib, desc, q, qd, sp, integ, nsub, np.float32)
    assert rc == 0 and np.abs(q2 - want[0]).max() < 2e-5 and np.abs(qd2 - want[1]).max() < 2e-5


@pytest.mark.parametrize("integ", [0, 1])
def test_a_robot_with_the_other_mirror_plane(hostmath_lib, integ):
    desc = _rotated_msj()
    oracle = TendonRobotOracle(desc)
    q, qd, sp = random_states(desc, 800, 32)
    want = oracle.step(q.astype(np.float64), qd.astype(np.float64), sp.astype(np.float64), integrator=integ)
    rc, q1, qd1, f1, mirror, half = _pairs(hostmath_lib, desc, q, qd, sp, integ)
    assert rc == 0 and mirror == 1                         # the y-z plane
    assert sorted(half) == list(range(8))
    assert np.abs(q1 - want[0]).max() < 1e-12 and np.abs(qd1 - want[1]).max() < 1e-11
    assert np.array_equal(f1, want[2])


def test_limit_hits_are_mirrored_too(msj_robot, msj_oracle, hostmath_lib):
    """States on and beyond the joint limits, large velocities: the clamp and the dropped outward velocity act on the
    mirrored env exactly as on the env."""
    desc = msj_robot.get_description()
    q, qd, sp = random_states(desc, 600, 33, vel_scale=1.0)
    q = (q * 1.05).astype(np.float32)
    sp[:] = -0.3
    want = msj_oracle.step(q.astype(np.float64), qd.astype(np.float64), sp.astype(np.float64), integrator=1)
    rc, q1, qd1, f1, _, _ = _pairs(hostmath_lib, desc, q, qd, sp, 1)
    assert rc == 0 and (~want[2]).sum() > 20
    assert np.abs(q1 - want[0]).max() < 1e-12 and np.abs(qd1 - want[1]).max() < 1e-11 and np.array_equal(f1, want[2])


def test_robots_without_a_mirror_plane_are_refused(hostmath_lib):
    from gym_roboy_amd.envs.robots import RobotDescription, msj_platform_spec
    from test_oracle import _random_ball_joint_robot
    desc = _random_ball_joint_robot(np.random.default_rng(3))
    q, qd, sp = random_states(desc, 4, 0)
    assert _pairs(hostmath_lib, desc, q, qd, sp, 0)[0] == 1
    # MsjRobot with one tendon moved, with an asymmetric joint limit, with the centre of mass off the axis: no plane either
    for edit in ("tendon", "limit", "com", "fmax"):
        spec = copy.deepcopy(msj_platform_spec())
        if edit == "tendon":
            spec["tendons"][2]["via_points"][2]["pos"][0] += 1e-3
        elif edit == "limit":
            spec["joints"][2]["limit"] = [-0.6, 0.5]
        elif edit == "com":
            spec["joints"][2]["com"] = [0.0, 0.002, 0.06]
        else:
            spec["tendons"][7]["f_max"] = 29.0
        d = RobotDescription(spec)
        q, qd, sp = random_states(d, 4, 0)
        assert _pairs(hostmath_lib, d, q, qd, sp, 0)[0] == 1, edit
