"""The two statements of the physics spec (numpy, C) agree on seeded random robots (tests/random_robots.py), so the
GPU parity tests on those robots (tests/test_random_robots_gpu.py) check the kernels against a twice-stated oracle."""
import numpy as np
import pytest

from conftest import random_states
from random_robots import random_tree_robot, random_tree_spec


@pytest.mark.parametrize("seed", range(16))
def test_c_and_numpy_oracles_agree_on_random_robots(seed):
    from oracle.c_oracle import COracle
    from oracle.physics_np import TendonRobotOracle
    robot, desc = random_tree_robot(seed)
    q, qd, sp = random_states(desc, 5, seed)
    integ = seed % 2
    a = TendonRobotOracle(desc).step(q.astype(np.float64), qd.astype(np.float64), sp.astype(np.float64), integrator=integ)
    b = COracle(desc, "f64").step(q, qd, sp, integrator=integ)
    assert np.abs(a[0] - b[0]).max() < 1e-12 and np.abs(a[1] - b[1]).max() < 1e-11
    assert np.array_equal(a[2], b[2])


def test_generator_is_deterministic_and_covers_the_shapes_it_promises():
    assert random_tree_spec(3) == random_tree_spec(3)
    roots = massless = stay = base = 0
    sizes = set()
    for seed in range(16):
        spec = random_tree_spec(seed)
        sizes.add(len(spec["joints"]))
        roots += sum(1 for j in spec["joints"] if j["parent"] < 0) > 1
        massless += any(j["mass"] == 0.0 for j in spec["joints"])
        for t in spec["tendons"]:
            links = [v["link"] for v in t["via_points"]]
            stay += len(set(links)) == 1
            base += -1 in links
    assert roots >= 3 and massless >= 8 and stay >= 1 and base >= 10 and min(sizes) <= 2 and max(sizes) >= 20


@pytest.mark.parametrize("seed", range(10))
def test_c_and_numpy_oracles_agree_on_random_ball_joint_robots(seed):
    from oracle.c_oracle import COracle
    from oracle.physics_np import TendonRobotOracle
    from random_robots import random_ball_joint_robot
    robot, desc = random_ball_joint_robot(seed, 8 if seed < 6 else 1 + 2 * seed)
    q, qd, sp = random_states(desc, 5, seed)
    integ = seed % 2
    a = TendonRobotOracle(desc).step(q.astype(np.float64), qd.astype(np.float64), sp.astype(np.float64), integrator=integ, n_substeps=1 + seed % 3)
    b = COracle(desc, "f64").step(q, qd, sp, integrator=integ, n_substeps=1 + seed % 3)
    assert np.abs(a[0] - b[0]).max() < 1e-12 and np.abs(a[1] - b[1]).max() < 1e-11
    assert np.array_equal(a[2], b[2])
