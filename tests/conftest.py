import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from build_dir import build_dir  # noqa: E402  (scratch builds live outside the repository)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _gpu_available():
    try:
        from gym_roboy_amd import _native
        return _native.device_count() > 0
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    # -m gpu on a box without a GPU must fail loudly, not skip: only skip GPU
    # tests when they were not asked for explicitly.
    if "gpu" in (config.getoption("-m") or ""):
        return
    if _gpu_available():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def msj_robot():
    from gym_roboy_amd.envs.robots import MsjRobot
    return MsjRobot()


@pytest.fixture(scope="session")
def msj_oracle(msj_robot):
    from oracle.physics_np import TendonRobotOracle
    return TendonRobotOracle(msj_robot.get_description())


@pytest.fixture(scope="session")
def hostmath_lib():
    """g++ build of the product's kernel arithmetic for the host (test harness)."""
    import ctypes
    build = build_dir()
    so = os.path.join(build, "libhostmath.so")
    src = os.path.join(ROOT, "tests", "hostmath", "host_math.cpp")
    deps = [src] + [os.path.join(ROOT, "gym_roboy_amd", "csrc", f) for f in ("msj_math.hpp", "msj_build.hpp")]
    if not os.path.exists(so) or any(os.path.getmtime(d) > os.path.getmtime(so) for d in deps):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
                               "-o", so, src])
    return ctypes.CDLL(so)


def random_states(desc, n, seed, vel_scale=1.0):
    """Seeded states inside the feasible region and set-points inside the action box."""
    rng = np.random.default_rng(seed)
    q = rng.uniform(0.98 * desc.q_lo, 0.98 * desc.q_hi, (n, desc.n_q))
    qd = rng.uniform(-desc.qd_max, desc.qd_max, (n, desc.n_q)) * vel_scale
    sp = rng.uniform(-0.3, 0.3, (n, desc.n_t))
    return q.astype(np.float32), qd.astype(np.float32), sp.astype(np.float32)
