"""The instrumented restatement behind the VALU roofline: the committed
count is what oracle/flop_count.cpp produces, and the instrumented run computes the same
step as the C oracle (a count taken along a wrong computation would be worthless)."""
import json

import numpy as np
import pytest

from gym_roboy_amd.envs.robots import MsjRobot
from oracle import flop_count as fc
from oracle.c_oracle import COracle


def test_committed_counts_are_reproduced():
    with open(fc.JSON_PATH) as fh:
        committed = json.load(fh)
    assert committed == fc.table()


@pytest.mark.parametrize("integ", [0, 1])
def test_instrumented_step_equals_the_oracle_step(integ):
    desc = MsjRobot().get_description()
    orc = COracle(desc, "f64")
    rng = np.random.default_rng(5)
    for _ in range(20):
        q = rng.uniform(0.8 * desc.q_lo, 0.8 * desc.q_hi)
        qd = rng.uniform(-desc.qd_max, desc.qd_max)
        sp = rng.uniform(-0.3, 0.3, desc.n_t)
        counts, q1, qd1, feas = fc.count_msj_step(desc, integ, q, qd, sp)
        qo, qdo, fo = orc.step(q[None], qd[None], sp[None], integrator=integ)
        assert np.abs(q1 - qo[0]).max() < 1e-11 and np.abs(qd1 - qdo[0]).max() < 1e-10
        assert feas == bool(fo[0])


def test_count_is_independent_of_the_data_and_scales_with_the_integrator():
    desc = MsjRobot().get_description()
    rng = np.random.default_rng(6)
    seen = {0: set(), 1: set()}
    for _ in range(8):
        q = rng.uniform(desc.q_lo, desc.q_hi)
        qd = rng.uniform(-desc.qd_max, desc.qd_max)
        sp = rng.uniform(-0.3, 0.3, desc.n_t)
        for integ in (0, 1):
            c = fc.count_msj_step(desc, integ, q, qd, sp)[0]
            # selections inside `limit` short-circuit on the host (the kernels evaluate both
            # sides): compare everything but that category
            seen[integ].add((c["add"], c["mul"], c["div"], c["trans"]))
    assert len(seen[0]) == 1 and len(seen[1]) == 1
    (e,), (r,) = seen[0], seen[1]
    assert r[3] == 4 * e[3]                    # four acceleration evaluations
    assert 3.9 < r[1] / e[1] < 4.2


@pytest.mark.parametrize("integ", [0, 1])
def test_instrumented_tree_step_equals_the_oracle_step(integ):
    """The scalar restatement of csrc/tree_aba.hpp's algorithm (articulated-body algorithm about the world
    origin, tendons as link crossings) against the C oracle (Jacobian-sum M, RNE bias, Cholesky) on the
    upper body and on MsjRobot."""
    from gym_roboy_amd.envs.robots import UpperBodyRobot
    for robot in (UpperBodyRobot(), MsjRobot()):
        desc = robot.get_description()
        orc = COracle(desc, "f64")
        rng = np.random.default_rng(7)
        for _ in range(6):
            q = rng.uniform(0.8 * desc.q_lo, 0.8 * desc.q_hi)
            qd = rng.uniform(-desc.qd_max, desc.qd_max)
            sp = rng.uniform(-0.3, 0.3, desc.n_t)
            counts, q1, qd1, feas = fc.count_tree_step(desc, integ, q, qd, sp)
            qo, qdo, fo = orc.step(q[None], qd[None], sp[None], integrator=integ)
            assert np.abs(q1 - qo[0]).max() < 1e-10 and np.abs(qd1 - qdo[0]).max() < 1e-9
            assert feas == bool(fo[0])
            assert counts["div"] == 0 and counts["trans"] == (1 if integ == 0 else 4) * (2 * desc.n_q + 4 * desc.n_t + desc.n_q)
