"""GPU parity of the generic joint-tree kernel on seeded RANDOM robots (tests/random_robots.py): random topology
(chains, bushes, several roots, up to 24 joints), random joint axes and inertial data with massless links, up to 40
tendons routed over random links.  Each robot: one env step of 37 envs from random states against the fp64 C oracle
(itself checked against the numpy statement of the spec on the same robots, tests/test_random_robots.py)."""
import numpy as np
import pytest

from conftest import random_states
from random_robots import random_tree_robot

pytestmark = pytest.mark.gpu
TOL = 2e-5          # fp32 kernel against the fp64 oracle, on top of which ...
REL_ACC = 3e-6      # ... of the env's largest joint acceleration times the step: a 32-link chain reaches 500 rad/s^2,
                    # where fp32's relative error through 32 levels of the recursion (measured 1.2e-6) exceeds TOL


def tolerance(desc, q, qd, sp, step=0.1):
    from oracle.physics_np import TendonRobotOracle
    acc = TendonRobotOracle(desc).acceleration(q.astype(np.float64), qd.astype(np.float64), sp.astype(np.float64))
    return TOL + REL_ACC * step * np.abs(acc).max(axis=1, keepdims=True)


# the kernel's limits (32 joints, 64 tendons), the deepest tree (a 32-link chain: 32 levels), the widest (a 31-child
# star: multi-pass levels, child lists beyond the 4 inline slots), the smallest
EXTREMES = {"max": dict(n_q=32, n_t=64), "chain32": dict(n_q=32, n_t=20, shape="chain"),
            "star32": dict(n_q=32, n_t=24, shape="star"), "one": dict(n_q=1, n_t=1), "one_many": dict(n_q=1, n_t=64)}


@pytest.mark.parametrize("seed", list(range(16)) + sorted(EXTREMES))
def test_random_tree_robot_matches_oracle(seed):
    from gym_roboy_amd.envs.simulations import HipBatchSimulation
    from oracle.c_oracle import COracle
    if isinstance(seed, str):
        robot, desc = random_tree_robot(100 + sorted(EXTREMES).index(seed), **EXTREMES[seed])
        seed = 100 + sorted(EXTREMES).index(seed)
    else:
        robot, desc = random_tree_robot(seed)
    integrator = "rk4" if seed % 2 else "euler"
    nsub = 2 if seed % 5 == 0 else 1
    n = 37
    q, qd, sp = random_states(desc, n, seed)
    sim = HipBatchSimulation(robot, n, integrator=integrator, n_substeps=nsub)
    assert sim.info()["kernel"] == 3
    sim.set_state(q, qd)
    q1, qd1, f1 = sim.forward_step_command(sp)
    qo, qdo, fo = COracle(desc, "f64").step(q, qd, sp, integrator=0 if integrator == "euler" else 1, n_substeps=nsub)
    tol = tolerance(desc, q, qd, sp)
    assert np.all(np.abs(q1 - qo) < tol), (desc.n_q, desc.n_t, np.abs(q1 - qo).max())
    assert np.all(np.abs(qd1 - qdo) < tol), (desc.n_q, desc.n_t, np.abs(qd1 - qdo).max())
    near = np.minimum(np.abs(qo - desc.q_lo), np.abs(qo - desc.q_hi)).min(axis=1) < 1e-5
    assert not np.any((f1 != fo) & ~near)
    # a second step from the kernel's own state: the working set of the first launch leaves nothing behind
    q2, qd2, _ = sim.forward_step_command(sp)
    qo2, qdo2, _ = COracle(desc, "f64").step(q1, qd1, sp, integrator=0 if integrator == "euler" else 1, n_substeps=nsub)
    tol = tolerance(desc, q1, qd1, sp)
    assert np.all(np.abs(q2 - qo2) < tol) and np.all(np.abs(qd2 - qdo2) < tol)
    sim.close()


@pytest.mark.parametrize("seed", [0, 3, 11, 13])
def test_random_tree_robot_env_per_lane_kernel_matches_oracle(seed):
    """The env-per-lane form, generated for the robot and built by hiprtc (an explicit choice: batches this small
    take the octets on their own), against the same oracle and against the octets."""
    from gym_roboy_amd.envs.simulations import HipBatchSimulation
    from oracle.c_oracle import COracle
    if isinstance(seed, str):
        robot, desc = random_tree_robot(100 + sorted(EXTREMES).index(seed), **EXTREMES[seed])
        seed = 100 + sorted(EXTREMES).index(seed)
    else:
        robot, desc = random_tree_robot(seed)
    integrator = "rk4" if seed % 2 else "euler"
    n = 70                                                     # one full wave and a ragged one
    q, qd, sp = random_states(desc, n, seed)
    sim = HipBatchSimulation(robot, n, integrator=integrator)
    sim.set_state(q, qd)
    qa, qda, fa = sim.forward_step_command(sp)                 # octets
    assert sim.info()["kernel"] == 3
    sim.select_kernel(1)
    assert sim.info()["kernel"] == 1 and sim.specialization() == "jit"
    sim.set_state(q, qd)
    q1, qd1, f1 = sim.forward_step_command(sp)
    qo, qdo, fo = COracle(desc, "f64").step(q, qd, sp, integrator=0 if integrator == "euler" else 1)
    tol = tolerance(desc, q, qd, sp)
    assert np.all(np.abs(q1 - qo) < tol), (desc.n_q, desc.n_t, np.abs(q1 - qo).max())
    assert np.all(np.abs(qd1 - qdo) < tol), (desc.n_q, desc.n_t, np.abs(qd1 - qdo).max())
    assert np.all(np.abs(q1 - qa) < 2 * tol) and np.all(np.abs(qd1 - qda) < 2 * tol)
    near = np.minimum(np.abs(qo - desc.q_lo), np.abs(qo - desc.q_hi)).min(axis=1) < 1e-5
    assert not np.any((f1 != fo) & ~near)
    sim.close()


@pytest.mark.parametrize("seed", [0, 2])
def test_robot_with_two_identical_branches_runs_them_as_pair_values(seed):
    """The env-per-lane form of a robot with two structurally identical branches (the generator writes them as one stream of
    pair values: v_pk_* instructions), built by hiprtc, against the oracle and the octets."""
    from gym_roboy_amd.envs.simulations import HipBatchSimulation
    from oracle.c_oracle import COracle
    from random_robots import random_mirrored_robot
    robot, desc = random_mirrored_robot(seed, n_branch=3 + seed, n_t_branch=3 + seed)
    integrator = "rk4" if seed else "euler"
    n = 70
    q, qd, sp = random_states(desc, n, seed)
    sim = HipBatchSimulation(robot, n, integrator=integrator)
    sim.set_state(q, qd)
    qa, qda, fa = sim.forward_step_command(sp)                 # octets
    sim.select_kernel(1)
    assert sim.info()["kernel"] == 1 and sim.specialization() == "jit"
    sim.set_state(q, qd)
    q1, qd1, f1 = sim.forward_step_command(sp)
    qo, qdo, fo = COracle(desc, "f64").step(q, qd, sp, integrator=0 if integrator == "euler" else 1)
    tol = tolerance(desc, q, qd, sp)
    assert np.all(np.abs(q1 - qo) < tol) and np.all(np.abs(qd1 - qdo) < tol), (np.abs(q1 - qo).max(), np.abs(qd1 - qdo).max())
    assert np.all(np.abs(q1 - qa) < 2 * tol) and np.all(np.abs(qd1 - qda) < 2 * tol)
    sim.close()


@pytest.mark.parametrize("kernel", [4, 6])      # the split form, and its lean two-part layout (round 5: parking in registers, exchange area over the row image)
@pytest.mark.parametrize("seed", [5, 9])        # 11 joints in 2 parts, 18 joints in 4 parts (several roots: no trunk; lean form: packed into 2)
def test_random_tree_robot_split_form_matches_oracle(seed, kernel):
    """The split forms (several waves per group of 64 envs) of a random robot, built by hiprtc on request."""
    from gym_roboy_amd.envs.simulations import HipBatchSimulation
    from oracle.c_oracle import COracle
    robot, desc = random_tree_robot(seed)
    integrator = "rk4" if seed % 2 else "euler"
    n = 130
    q, qd, sp = random_states(desc, n, seed)
    sim = HipBatchSimulation(robot, n, integrator=integrator, n_substeps=2 if seed == 5 else 1)
    sim.select_kernel(kernel)
    assert sim.info()["kernel"] == kernel and sim.specialization() == "jit"
    sim.set_state(q, qd)
    q1, qd1, f1 = sim.forward_step_command(sp)
    qo, qdo, fo = COracle(desc, "f64").step(q, qd, sp, integrator=0 if integrator == "euler" else 1, n_substeps=2 if seed == 5 else 1)
    tol = tolerance(desc, q, qd, sp)
    assert np.all(np.abs(q1 - qo) < tol), (desc.n_q, desc.n_t, np.abs(q1 - qo).max())
    assert np.all(np.abs(qd1 - qdo) < tol), (desc.n_q, desc.n_t, np.abs(qd1 - qdo).max())
    near = np.minimum(np.abs(qo - desc.q_lo), np.abs(qo - desc.q_hi)).min(axis=1) < 1e-5
    assert not np.any((f1 != fo) & ~near)
    sim.close()


def _cache_stats():
    import ctypes
    from gym_roboy_amd import _native
    v = [ctypes.c_int64(0) for _ in range(3)]
    _native.load().rb_jit_cache_stats(*[ctypes.byref(x) for x in v])
    return tuple(x.value for x in v)                           # hits, compiles, stores


def test_hiprtc_code_objects_are_cached_on_disk_and_a_damaged_file_is_rebuilt(tmp_path, monkeypatch):
    """ROBOY_SIM_JIT_CACHE: the first handle compiles and stores, the second loads the stored code object (same results bit
    for bit), a truncated file is ignored and replaced, and "0" switches the cache off."""
    from gym_roboy_amd.envs.simulations import HipBatchSimulation
    monkeypatch.setenv("ROBOY_SIM_JIT_CACHE", str(tmp_path))
    robot, desc = random_tree_robot(3)
    n = 70
    q, qd, sp = random_states(desc, n, 3)

    def run():
        sim = HipBatchSimulation(robot, n)
        sim.select_kernel(1)
        assert sim.specialization() == "jit"
        sim.set_state(q, qd)
        out = sim.forward_step_command(sp)
        sim.close()
        return out

    h0, c0, s0 = _cache_stats()
    first = run()
    h1, c1, s1 = _cache_stats()
    files = sorted(tmp_path.glob("*.rbjc"))
    assert (h1 - h0, c1 - c0, s1 - s0) == (0, 1, 1) and len(files) == 1 and files[0].stat().st_size > 10000
    second = run()
    h2, c2, s2 = _cache_stats()
    assert (h2 - h1, c2 - c1, s2 - s1) == (1, 0, 0)
    assert all(np.array_equal(a, b) for a, b in zip(first, second))
    data = files[0].read_bytes()
    files[0].write_bytes(data[:len(data) // 2])                 # a half-written or damaged file: ignored, rebuilt, replaced
    third = run()
    h3, c3, s3 = _cache_stats()
    assert (h3 - h2, c3 - c2, s3 - s2) == (0, 1, 1) and files[0].read_bytes() == data
    assert all(np.array_equal(a, b) for a, b in zip(first, third))
    monkeypatch.setenv("ROBOY_SIM_JIT_CACHE", "0")
    run()
    h4, c4, s4 = _cache_stats()
    assert (h4 - h3, c4 - c3, s4 - s3) == (0, 1, 0)
    assert not list(tmp_path.glob("*.tmp.*"))


def test_split_form_is_refused_where_the_tree_has_no_parts():
    from gym_roboy_amd.envs.simulations import HipBatchSimulation
    robot, _ = random_tree_robot(41, n_q=6, n_t=3, shape="chain")
    sim = HipBatchSimulation(robot, 10)
    for k in (4, 6):
        with pytest.raises(Exception, match="no split form"):
            sim.select_kernel(k)
    sim.close()


def test_library_builds_the_lane_kernels_on_its_own_only_where_they_fit():
    """The library's own choice at 16 384 envs: a robot whose generated code keeps few values alive gets the hiprtc-built
    env-per-lane kernels (and they match the oracle on a sample); a dense 22-joint robot whose live set exceeds a SIMD's
    register file keeps the octets (no minutes-long build of a kernel that would spill)."""
    from gym_roboy_amd.envs.simulations import HipBatchSimulation
    from oracle.c_oracle import COracle
    n = 16384
    robot, desc = random_tree_robot(3)                       # 8 joints, 4 tendons
    q, qd, sp = random_states(desc, n, 3)
    sim = HipBatchSimulation(robot, n, integrator="euler")
    sim.set_state(q, qd)
    q1, qd1, f1 = sim.forward_step_command(sp)
    assert sim.info()["kernel"] == 1 and sim.specialization() == "jit"
    idx = np.arange(0, n, 97)
    qo, qdo, fo = COracle(desc, "f64").step(q[idx], qd[idx], sp[idx], integrator=0)
    tol = tolerance(desc, q[idx], qd[idx], sp[idx])
    assert np.all(np.abs(q1[idx] - qo) < tol) and np.all(np.abs(qd1[idx] - qdo) < tol)
    sim.close()
    robot, desc = random_tree_robot(1)                       # 22 joints, 24 tendons, dense: ~740 values alive at once
    sim = HipBatchSimulation(robot, n, integrator="euler")
    q, qd, sp = random_states(desc, n, 1)
    sim.set_state(q, qd)
    sim.forward_step_command(sp)
    assert sim.info()["kernel"] == 3 and sim.specialization() == "kernarg"
    sim.close()


def test_random_robot_fused_env_layer_runs_and_matches_plain_step():
    """The fused env kernel of the joint-tree class on a random robot: its physics leg equals the plain step."""
    from gym_roboy_amd.envs.simulations import HipBatchSimulation
    from gym_roboy_amd.envs.vec_env import RoboyVecEnv
    robot, desc = random_tree_robot(6)
    n = 129
    env = RoboyVecEnv(robot, n, seed=3, auto_reset=False)
    obs0 = env.reset()
    rng = np.random.default_rng(0)
    act = rng.uniform(-1, 1, (n, desc.n_t)).astype(np.float32)
    obs, rew, done, _ = env.step(act)
    obs = np.asarray(obs.cpu() if hasattr(obs, "cpu") else obs)
    sim = HipBatchSimulation(robot, n)
    sim.set_state(np.zeros((n, desc.n_q), np.float32), np.zeros((n, desc.n_q), np.float32))
    hi = float(robot.get_action_space().high[0])
    q1, qd1, _ = sim.forward_step_command(act * hi)
    assert np.abs(obs[:, :desc.n_q] - q1).max() < 1e-6 and np.abs(obs[:, desc.n_q:2 * desc.n_q] - qd1).max() < 1e-6
    assert np.all(np.isfinite(np.asarray(rew.cpu() if hasattr(rew, "cpu") else rew)))
    sim.close()


# ---- the ball-joint class (closed form, msj_math.hpp) on random robots of that class, through every kernel form ----
def _ball_check(robot, desc, n, integrator, nsub, seed, kernel=0, sample=1):
    from gym_roboy_amd.envs.simulations import HipBatchSimulation
    from oracle.c_oracle import COracle
    q, qd, sp = random_states(desc, n, seed)
    sim = HipBatchSimulation(robot, n, integrator=integrator, n_substeps=nsub)
    if kernel:
        sim.select_kernel(kernel)
    assert sim.info()["kernel"] in (1, 2)                      # never the joint-tree kernel
    sim.set_state(q, qd)
    q1, qd1, f1 = sim.forward_step_command(sp)
    idx = np.arange(0, n, sample)
    qo, qdo, fo = COracle(desc, "f64").step(q[idx], qd[idx], sp[idx], integrator=0 if integrator == "euler" else 1, n_substeps=nsub)
    assert np.abs(q1[idx] - qo).max() < TOL, np.abs(q1[idx] - qo).max()
    assert np.abs(qd1[idx] - qdo).max() < TOL, np.abs(qd1[idx] - qdo).max()
    near = np.minimum(np.abs(qo - desc.q_lo), np.abs(qo - desc.q_hi)).min(axis=1) < 1e-5
    assert not np.any((f1[idx] != fo) & ~near)
    spec = sim.specialization()
    sim.close()
    return spec


@pytest.mark.parametrize("seed", range(10))
def test_random_ball_joint_robot_small_batches(seed):
    """8 tendons: tendon-per-lane (AUTO below 8 192 envs) and env-per-lane forms; both integrators; 1-3 substeps."""
    from random_robots import random_ball_joint_robot
    robot, desc = random_ball_joint_robot(seed, 8)
    integrator = "rk4" if seed % 2 else "euler"
    nsub = 1 + seed % 3
    _ball_check(robot, desc, 257, integrator, nsub, seed)                 # AUTO: tendon per lane
    _ball_check(robot, desc, 257, integrator, nsub, seed, kernel=1)       # env per lane, one wave per workgroup
    _ball_check(robot, desc, 9000, integrator, nsub, seed, sample=7)


@pytest.mark.parametrize("n_t", [1, 2, 5, 7, 9, 13, 16])
def test_random_ball_joint_robot_other_tendon_counts(n_t):
    """Run-time tendon count (the NTX kernels), small and large launch configuration."""
    from random_robots import random_ball_joint_robot
    robot, desc = random_ball_joint_robot(20 + n_t, n_t)
    _ball_check(robot, desc, 515, "rk4" if n_t % 2 else "euler", 1 + n_t % 2, n_t)
    _ball_check(robot, desc, 70001, "euler" if n_t % 2 else "rk4", 1, n_t, sample=53)


@pytest.mark.parametrize("seed", [1, 3, 6])
def test_random_ball_joint_robot_large_batches_jit_and_kernarg(seed):
    """Above 65 536 envs: the hiprtc instances on the robot's own constants and, with ROBOY_SIM_JIT=0, the kernarg
    instances (seed 3 and 6: 2 and 1 substeps; seed 3 takes the principal-axis branch)."""
    import os
    from random_robots import random_ball_joint_robot
    robot, desc = random_ball_joint_robot(seed, 8)
    integrator = "rk4" if seed % 2 else "euler"
    assert _ball_check(robot, desc, 70001, integrator, 1 + seed % 2, seed, sample=41) == "jit"
    os.environ["ROBOY_SIM_JIT"] = "0"
    try:
        assert _ball_check(robot, desc, 70001, integrator, 1 + seed % 2, seed, sample=41) == "kernarg"
    finally:
        del os.environ["ROBOY_SIM_JIT"]


@pytest.mark.parametrize("seed", list(range(16)) + sorted(EXTREMES))
def test_random_tree_robots_stay_finite_and_inside_their_boxes_under_extreme_actions(seed):
    """150 steps of bang-bang set-points from the corners of the state box (both integrators): every state stays
    finite, inside the joint limits and the velocity box - whatever the topology (the 0 * inf of drifting idle lanes
    showed up only on deep chains and only for some poses)."""
    from gym_roboy_amd.envs.simulations import HipBatchSimulation
    if isinstance(seed, str):
        robot, desc = random_tree_robot(100 + sorted(EXTREMES).index(seed), **EXTREMES[seed])
        seed = 100 + sorted(EXTREMES).index(seed)
    else:
        robot, desc = random_tree_robot(seed)
    n = 64
    rng = np.random.default_rng(seed)
    for integrator in ("euler", "rk4"):
        sim = HipBatchSimulation(robot, n, integrator=integrator)
        corner = rng.integers(0, 2, (n, desc.n_q)).astype(bool)
        q0 = np.where(corner, desc.q_hi, desc.q_lo).astype(np.float32) * 0.999
        qd0 = np.where(rng.integers(0, 2, (n, desc.n_q)).astype(bool), desc.qd_max, -desc.qd_max).astype(np.float32)
        sim.set_state(q0, qd0)
        for t in range(150):
            sp = (0.3 * np.sign(rng.normal(size=(n, desc.n_t)))).astype(np.float32)
            q, qd, f = sim.forward_step_command(sp)
            if t % 50 == 49 or t == 0:
                assert np.isfinite(q).all() and np.isfinite(qd).all(), (integrator, t)
                assert np.all(q >= desc.q_lo - 1e-6) and np.all(q <= desc.q_hi + 1e-6), (integrator, t)
                assert np.all(np.abs(qd) <= desc.qd_max + 1e-6), (integrator, t)
        sim.close()


@pytest.mark.parametrize("seed,n_t", [(0, 8), (1, 8), (2, 8), (3, 8), (4, 1), (5, 5), (6, 13), (7, 16)])
def test_random_ball_joint_robots_stay_finite_and_inside_their_boxes_under_extreme_actions(seed, n_t):
    from gym_roboy_amd.envs.simulations import HipBatchSimulation
    from random_robots import random_ball_joint_robot
    robot, desc = random_ball_joint_robot(seed, n_t)
    rng = np.random.default_rng(seed)
    for integrator, n in (("euler", 300), ("rk4", 300), ("euler", 70001)):
        sim = HipBatchSimulation(robot, n, integrator=integrator, n_substeps=1 + seed % 2)
        q0 = np.where(rng.integers(0, 2, (n, 3)).astype(bool), desc.q_hi, desc.q_lo).astype(np.float32) * 0.999
        qd0 = np.where(rng.integers(0, 2, (n, 3)).astype(bool), desc.qd_max, -desc.qd_max).astype(np.float32)
        sim.set_state(q0, qd0)
        for t in range(60 if n > 1000 else 150):
            sp = (0.3 * np.sign(rng.normal(size=(n, n_t)))).astype(np.float32)
            q, qd, f = sim.forward_step_command(sp)
        assert np.isfinite(q).all() and np.isfinite(qd).all(), (integrator, n)
        assert np.all(q >= desc.q_lo - 1e-6) and np.all(q <= desc.q_hi + 1e-6)
        assert np.all(np.abs(qd) <= desc.qd_max + 1e-6)
        sim.close()


@pytest.mark.parametrize("seed", [1, 2, 6, "star32", "chain32"])
def test_fused_env_layer_on_random_robots_matches_host_replay(seed):
    """RoboyVecEnv over the joint-tree kernel on random robots: the replay check of tests/test_tree_robot_gpu.py
    (states and goals bit for bit against the plain kernel + numpy Philox, reward / done against reward.py in float64,
    episode statistics), with auto-reset and a 9-step episode horizon so that every bookkeeping path runs."""
    from host_env_model import HipStepper, HostEnvModel
    from gym_roboy_amd.envs.vec_env import RoboyVecEnv
    if isinstance(seed, str):
        robot, desc = random_tree_robot(100 + sorted(EXTREMES).index(seed), **EXTREMES[seed])
    else:
        robot, desc = random_tree_robot(seed)
    n, env_seed, max_len = 66, 3, 9
    vec = RoboyVecEnv(robot, n, seed=env_seed, auto_reset=True, max_episode_length=max_len, joint_vel_penalty=True)
    host = HostEnvModel(robot, HipStepper(robot, n, env_seed), n, env_seed, max_len, True, True, True)
    obs0 = vec.reset()
    host.goal = host.draw(np.ones(n, bool))
    nq = desc.n_q
    assert obs0.shape == (n, 3 * nq) and not obs0[:, :2 * nq].any() and np.array_equal(obs0[:, 2 * nq:], host.goal)
    rng = np.random.default_rng(1)
    n_done = 0
    for t in range(22):
        a = rng.uniform(-1, 1, (n, desc.n_t)).astype(np.float32)
        obs, rew, done, _ = vec.step(a)
        h_obs, h_rew, h_done, margin = host.step(a)
        assert np.array_equal(done, h_done) or (margin[done != h_done] < 1e-5).all()
        assert (done == h_done).all()
        assert np.array_equal(obs, h_obs.astype(np.float32))
        np.testing.assert_allclose(rew, h_rew, rtol=3e-5, atol=3e-4)
        n_done += int(done.sum())
    assert n_done >= 2 * n
    st = vec.stats()
    assert st["n_env_steps"] == 22 * n and st["n_episodes"] == n_done
    vec.close(); host.stepper.close()


@pytest.mark.gpu
@pytest.mark.parametrize("kernel", [1, 4, 6])       # RB_KERNEL_ENV_PER_LANE, _SPLIT, _SPLIT2 (the lean two-part layout): all built by hiprtc for this robot
def test_fused_env_layer_of_the_hiprtc_built_lane_and_split_kernels_matches_host_replay(kernel):
    """The env-step kernels of the env-per-lane forms take ONE struct argument (env_common.hpp: TreeEnvArgs) and read most of it
    behind the step; for a robot without ahead-of-time instances they are compiled by hiprtc and launched through
    hipModuleLaunchKernel with that struct: the replay check of the test above on those two forms (random robot 4 has a split form
    with tendon helpers), the host model stepping with the same kernel form.  The plain step and the fused env step of a form are
    separate compilations of the same generated text; the text writes its fused multiply-adds out (rbl_fma), so the two agree bit
    for bit - while a*b + c*d was left to the compiler's contraction the one-wave form of this robot differed by 6e-7 after two
    steps."""
    from host_env_model import HipStepper, HostEnvModel
    from gym_roboy_amd.envs.vec_env import RoboyVecEnv
    robot, desc = random_tree_robot(4)
    n, env_seed, max_len = 130, 5, 7
    vec = RoboyVecEnv(robot, n, seed=env_seed, auto_reset=True, max_episode_length=max_len, joint_vel_penalty=True)
    vec.sim.select_kernel(kernel)
    stepper = HipStepper(robot, n, env_seed)
    stepper.sim.select_kernel(kernel)
    host = HostEnvModel(robot, stepper, n, env_seed, max_len, True, True, True)
    obs0 = vec.reset()
    host.goal = host.draw(np.ones(n, bool))
    nq = desc.n_q
    assert np.array_equal(obs0[:, 2 * nq:], host.goal)
    rng = np.random.default_rng(2)
    n_done = 0
    for t in range(16):
        a = rng.uniform(-1, 1, (n, desc.n_t)).astype(np.float32)
        obs, rew, done, _ = vec.step(a)
        h_obs, h_rew, h_done, margin = host.step(a)
        assert (done == h_done).all() or (margin[done != h_done] < 1e-5).all()
        assert np.array_equal(obs, h_obs.astype(np.float32))
        np.testing.assert_allclose(rew, h_rew, rtol=3e-5, atol=3e-4)
        n_done += int(done.sum())
    assert n_done >= 2 * n
    assert vec.sim.info()["kernel"] == kernel
    st = vec.stats()
    assert st["n_env_steps"] == 16 * n and st["n_episodes"] == n_done
    vec.close(); stepper.close()
