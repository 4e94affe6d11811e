"""GPU parity: the HIP physics step (through the C ABI) against the fp64 oracle.

Tolerance: BASELINE.json's north_star asks for per-step state error < 1e-4 in
fp32; the kernels are held to 2e-5 here (observed ~2e-6).  The feasibility
flag is a discontinuity, so flag mismatches are only accepted for envs whose
oracle state sits within 1e-5 of a joint limit.
"""
import numpy as np
import pytest

from conftest import random_states

pytestmark = pytest.mark.gpu

TOL = 2e-5


def _sim(robot, n, **kw):
    from gym_roboy_amd.envs.simulations import HipBatchSimulation
    return HipBatchSimulation(robot, n, **kw)


KERNELS = {"env_per_lane": 1, "tendon_per_lane": 2, "lane_pair": 5}


def _auto_kernel(robot, n, integrator):
    """What RB_KERNEL_AUTO picks for the plain step of a ball-joint robot - evaluated from the rules the library exports
    (rb_auto_rules; csrc/roboy_dispatch.hpp: AUTO_RULES), not restated here."""
    from gym_roboy_amd import _native as nat
    if robot.get_description().n_t != 8:
        return 1
    mirror = type(robot).__name__ in ("MsjRobot", "Turned")
    return nat.auto_kernel(0, 0, 0 if integrator == "euler" else 1, nat.RB_NEED_MIRROR if mirror else nat.RB_NEED_NO_MIRROR, n)


def _check_step(robot, oracle, n, integrator, nsub, seed, kernel=0):
    from oracle.physics_np import EULER, RK4
    desc = robot.get_description()
    q, qd, sp = random_states(desc, n, seed)
    sim = _sim(robot, n, integrator=integrator, n_substeps=nsub)
    sim.select_kernel(kernel)
    assert sim.info()["kernel"] == (kernel or _auto_kernel(robot, n, integrator))
    sim.set_state(q, qd)
    q1, qd1, f1 = sim.forward_step_command(sp)
    qo, qdo, fo = oracle.step(q.astype(np.float64), qd.astype(np.float64), sp.astype(np.float64),
                              integrator=EULER if integrator == "euler" else RK4, n_substeps=nsub)
    assert np.abs(q1 - qo).max() < TOL
    assert np.abs(qd1 - qdo).max() < TOL
    near = (np.minimum(np.abs(qo - desc.q_lo), np.abs(qo - desc.q_hi)).min(axis=1) < 1e-5)
    mismatch = f1 != fo
    assert not np.any(mismatch & ~near)
    sim.close()
    return np.abs(q1 - qo).max(), np.abs(qd1 - qdo).max()


@pytest.mark.parametrize("kernel", sorted(KERNELS))
@pytest.mark.parametrize("integrator", ["euler", "rk4"])
@pytest.mark.parametrize("nsub", [1, 4])
@pytest.mark.parametrize("n", [1, 63, 4097])
def test_step_matches_oracle(msj_robot, msj_oracle, kernel, integrator, nsub, n):
    """Both kernel forms, ragged sizes (1 env; 63 = not a whole wave; 4097 = one
    env into the next wave / the next 8-lane group)."""
    _check_step(msj_robot, msj_oracle, n, integrator, nsub, seed=n + nsub, kernel=KERNELS[kernel])


@pytest.mark.parametrize("integrator", ["euler", "rk4"])
def test_step_large_batch_matches_oracle(msj_robot, msj_oracle, integrator):
    # > 65536 envs takes the rolled-loop, 256-thread launch configuration (AUTO)
    _check_step(msj_robot, msj_oracle, 70001, integrator, 1, seed=5)


@pytest.mark.parametrize("n", [4096, 4097, 12288, 12289, 16384, 16385, 32768, 32769, 65536, 65537])
def test_step_at_the_dispatch_boundaries_matches_oracle(msj_robot, n):
    """Batch sizes on either side of every kernel-form switch of AUTO (MsjRobot: tendon per lane up to 4 096 / 12 288 envs for
    Euler / RK4, two lanes per env up to 16 384 / 32 768, then one env per lane - one-wave workgroups up to 65 536, 256-thread
    workgroups above): a strided sample against the C oracle, the last env included."""
    from gym_roboy_amd.envs.simulations import HipBatchSimulation
    from oracle.c_oracle import COracle
    desc = msj_robot.get_description()
    q, qd, sp = random_states(desc, n, n)
    idx = np.unique(np.concatenate([np.arange(0, n, 61), [n - 1]]))
    orc = COracle(desc, "f64")
    for integrator in ("euler", "rk4"):
        sim = HipBatchSimulation(msj_robot, n, integrator=integrator)
        assert sim.info()["kernel"] == _auto_kernel(msj_robot, n, integrator)
        sim.set_state(q, qd)
        q1, qd1, f1 = sim.forward_step_command(sp)
        qo, qdo, fo = orc.step(q[idx], qd[idx], sp[idx], integrator=0 if integrator == "euler" else 1)
        assert np.abs(q1[idx] - qo).max() < TOL and np.abs(qd1[idx] - qdo).max() < TOL
        sim.close()


def test_kernel_forms_agree_with_each_other(msj_robot):
    """Same inputs through all three forms: they differ only in the order the 8
    tendon torques are summed (sequential / DPP butterfly / two halves) and, for the
    pair form, in RK4's summation order."""
    n = 3000
    q, qd, sp = random_states(msj_robot.get_description(), n, 21)
    out = []
    for kernel in (1, 2, 5):
        sim = _sim(msj_robot, n, integrator="rk4")
        sim.select_kernel(kernel)
        sim.set_state(q, qd)
        out.append(sim.forward_step_command(sp))
        sim.close()
    for other in out[1:]:
        assert np.abs(out[0][0] - other[0]).max() < 5e-6 and np.abs(out[0][1] - other[1]).max() < 5e-6
        assert np.mean(out[0][2] == other[2]) > 0.999


@pytest.mark.parametrize("integrator", ["euler", "rk4"])
@pytest.mark.parametrize("n", [40000, 131072 + 77])
def test_pair_form_with_256_thread_workgroups_matches_oracle(msj_robot, integrator, n):
    """The two-lanes-per-env form above its one-wave-workgroup range (ragged last workgroup): a strided sample against
    the C oracle, the last env included."""
    from oracle.c_oracle import COracle
    desc = msj_robot.get_description()
    q, qd, sp = random_states(desc, n, n)
    idx = np.unique(np.concatenate([np.arange(0, n, 53), [n - 2, n - 1]]))
    sim = _sim(msj_robot, n, integrator=integrator)
    sim.select_kernel(KERNELS["lane_pair"])
    assert sim.info()["kernel"] == KERNELS["lane_pair"]
    sim.set_state(q, qd)
    q1, qd1, f1 = sim.forward_step_command(sp)
    qo, qdo, fo = COracle(desc, "f64").step(q[idx], qd[idx], sp[idx], integrator=0 if integrator == "euler" else 1)
    assert np.abs(q1[idx] - qo).max() < TOL and np.abs(qd1[idx] - qdo).max() < TOL
    sim.close()


@pytest.mark.parametrize("integrator", ["euler", "rk4"])
def test_pair_form_on_a_robot_with_the_other_mirror_plane_and_on_kernarg_constants(msj_robot, integrator):
    """MsjRobot turned by 90 degrees about z: its mirror plane is the y-z plane, its constants are not the baked table's
    (kernarg instances); and robots without a mirror plane refuse the form and leave the handle as it was."""
    from oracle.c_oracle import COracle
    from test_mirror_pairs import _rotated_msj
    desc = _rotated_msj()

    class Turned(type(msj_robot)):
        @classmethod
        def get_description(cls):
            return desc
    robot = Turned()
    _check_step(robot, COracle(desc, "f64"), 5000, integrator, 2, seed=77, kernel=KERNELS["lane_pair"])
    skew = _ball_joint_robot(msj_robot, 8, 4)
    sim = _sim(skew, 64, integrator=integrator)
    before = sim.info()["kernel"]
    with pytest.raises(Exception, match="mirror plane"):
        sim.select_kernel(KERNELS["lane_pair"])
    assert sim.info()["kernel"] == before
    sim.close()


def test_reset_gives_zero_state(msj_robot):
    sim = _sim(msj_robot, 100)
    q, qd, sp = random_states(msj_robot.get_description(), 100, 0)
    sim.set_state(q, qd)
    q0, qd0, f0 = sim.forward_reset_command()
    assert np.all(q0 == 0) and np.all(qd0 == 0) and np.all(f0)
    # masked reset only touches the selected envs
    sim.set_state(q, qd)
    mask = np.arange(100) % 2 == 0
    q1, qd1, _ = sim.forward_reset_command(mask)
    assert np.all(q1[mask] == 0) and np.array_equal(q1[~mask], q[~mask])
    sim.close()


def test_zero_action_at_rest_is_equilibrium(msj_robot):
    sim = _sim(msj_robot, 8)
    for _ in range(5):
        q, qd, f = sim.forward_step_command(np.zeros((8, 8), np.float32))
    assert np.all(q == 0) and np.all(qd == 0) and np.all(f)
    sim.close()


def test_read_state_is_idempotent(msj_robot):
    sim = _sim(msj_robot, 16)
    q, qd, sp = random_states(msj_robot.get_description(), 16, 1)
    sim.set_state(q, qd)
    a = sim.read_state()
    b = sim.read_state()
    for x, y in zip(a, b):
        assert np.array_equal(x, y)
    assert np.array_equal(a[0], q) and np.array_equal(a[1], qd)
    sim.close()


@pytest.mark.parametrize("integrator", ["euler", "rk4"])
def test_rollout_tracks_oracle_over_an_episode(msj_robot, msj_oracle, integrator):
    """400 steps (one episode, roboy_env.py:28) of held-then-changed set-points:
    the fp32 trajectory must stay close to the fp64 one (contractive dynamics)."""
    from oracle.physics_np import EULER, RK4
    n = 256
    rng = np.random.default_rng(11)
    sim = _sim(msj_robot, n, integrator=integrator)
    qo = np.zeros((n, 3)); qdo = np.zeros((n, 3))
    worst = 0.0
    for t in range(400):
        if t % 25 == 0:
            sp = rng.uniform(-0.3, 0.3, (n, 8)).astype(np.float32)
        q1, qd1, f1 = sim.forward_step_command(sp)
        qo, qdo, fo = msj_oracle.step(qo, qdo, sp.astype(np.float64),
                                      integrator=EULER if integrator == "euler" else RK4)
        worst = max(worst, np.abs(q1 - qo).max(), np.abs(qd1 - qdo).max())
    assert worst < 5e-4, worst
    sim.close()


@pytest.mark.parametrize("steps", [100, 403])     # 403 = 3 graphs of 128 + one of 16 + 3 eager steps
def test_rollout_dev_equals_repeated_step_dev(msj_robot, steps):
    """The rollout entry point (eager and hipGraph) is bit-identical to
    single-step launches over the same action ring."""
    n, ring = 4096, 4
    sims = [_sim(msj_robot, n, seed=7) for _ in range(3)]
    outs = []
    for mode, sim in enumerate(sims):
        d_ring = sim.malloc(4 * ring * n * 8)
        for r in range(ring):
            sim.fill_actions_dev(d_ring + 4 * r * n * 8, r)
        if mode == 0:
            for t in range(steps):
                sim.step_dev(d_ring + 4 * (t % ring) * n * 8, 0.3)
        else:
            sim.rollout_dev(d_ring, ring, steps, 0.3, use_graph=(mode == 2))
        sim.synchronize()
        outs.append(sim.read_state())
    for o in outs[1:]:
        assert np.array_equal(o[0], outs[0][0]) and np.array_equal(o[1], outs[0][1])
        assert np.array_equal(o[2], outs[0][2])
    assert np.abs(outs[0][0]).max() > 0.01   # it actually moved
    for s in sims:
        s.close()


@pytest.mark.parametrize("integrator", ["euler", "rk4"])
@pytest.mark.parametrize("n", [4096, 100001])
def test_fused_open_loop_rollout_equals_single_steps(msj_robot, integrator, n):
    """rb_rollout_fused_dev (state in registers across steps, one launch) is
    bit-identical to the same number of rb_step_dev launches, for both batch
    regimes of the env-per-lane kernel, including the last step's feasibility."""
    ring, steps = 3, 41
    sims = [_sim(msj_robot, n, seed=11, integrator=integrator, n_substeps=2) for _ in range(2)]
    outs = []
    for mode, sim in enumerate(sims):
        sim.select_kernel(KERNELS["env_per_lane"])
        d_ring = sim.malloc(4 * ring * n * 8)
        for r in range(ring):
            sim.fill_actions_dev(d_ring + 4 * r * n * 8, r)
        if mode == 0:
            for t in range(steps):
                sim.step_dev(d_ring + 4 * (t % ring) * n * 8, 0.9)
        else:
            sim.rollout_fused_dev(d_ring, ring, steps, 0.9)
            sim.rollout_fused_dev(d_ring, ring, 0, 0.9)     # zero steps: a no-op
        sim.synchronize()
        outs.append(sim.read_state())
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])
    assert np.array_equal(outs[0][2], outs[1][2])
    assert 0 < outs[0][2].sum() < n       # some envs ended infeasible, some not
    for s in sims:
        s.close()


def test_fused_rollout_is_refused_for_tree_robots():
    from gym_roboy_amd.envs.robots import UpperBodyRobot
    sim = _sim(UpperBodyRobot(), 8)
    d = sim.malloc(4 * 8 * sim.n_t)
    with pytest.raises(Exception, match="ball-joint"):
        sim.rollout_fused_dev(d, 1, 2, 1.0)
    sim.close()


def test_stream_selection_own_default_and_given(msj_robot):
    """set_stream(None) = the handle's own stream, 0 = the device's default (null) stream (what a
    framework's current stream is by default; not the same thing as "own"), else a hipStream_t.
    The three give the same results; graph capture is refused on the null stream, loudly."""
    import torch
    n = 4096
    outs = []
    for which in ("own", "default", "given"):
        sim = _sim(msj_robot, n, seed=2)
        side = torch.cuda.Stream()
        sim.set_stream({"own": None, "default": 0, "given": side.cuda_stream}[which])
        d_act = sim.malloc(4 * n * 8)
        sim.fill_actions_dev(d_act, 0)
        for _ in range(5):
            sim.step_dev(d_act, 0.3)
        if which == "default":
            with pytest.raises(ValueError, match="non-default stream"):
                sim.rollout_dev(d_act, 1, 16, 0.3, use_graph=True)
            torch.cuda.synchronize()          # a torch-side sync of the default stream covers the launches
        else:
            sim.rollout_dev(d_act, 1, 16, 0.3, use_graph=True)
            sim.rollout_dev(d_act, 1, 16, 0.3, use_graph=False)
            sim.synchronize()
        outs.append(sim.read_state())
        sim.close()
    assert np.array_equal(outs[0][0], outs[2][0]) and np.array_equal(outs[0][1], outs[2][1])
    assert np.abs(outs[1][0]).max() > 0          # the default-stream steps ran


def test_device_pointer_accessors(msj_robot):
    """rb_state_ptrs exposes the SoA planes q[j][N], qd[j][N], feasible[N] in place;
    rb_sample_goals_dev draws the same goals as the host entry point; rb_env_stats_dev
    delivers the same block as rb_env_stats."""
    import ctypes
    from gym_roboy_amd import _native as nat
    from gym_roboy_amd.envs import RoboyVecEnv
    n = 1000
    a, b = _sim(msj_robot, n, seed=8), _sim(msj_robot, n, seed=8)
    desc = msj_robot.get_description()
    q, qd, sp = random_states(desc, n, 21)
    a.set_state(q, qd)
    a.forward_step_command(sp)
    dq, dqd, dfeas = a.state_ptrs()
    planes_q = a.download(dq, (3, n))               # [j][N]
    planes_qd = a.download(dqd, (3, n))
    feas = a.download(dfeas, (n,), np.uint32)
    rq, rqd, rf = a.read_state()
    assert np.array_equal(planes_q.T, rq) and np.array_equal(planes_qd.T, rqd) and np.array_equal(feas != 0, rf)
    # goals: device planes [j][N] vs host rows [N][j], same draw index on two handles with the same seed
    d_goal = a.malloc(4 * 3 * n)
    a.sample_goals_dev(d_goal)
    a.synchronize()
    assert np.array_equal(a.download(d_goal, (3, n)).T, b.get_new_goal_joint_angles())
    mask = (np.arange(n) % 3 == 0).astype(np.uint8)
    d_mask = a.malloc(n)
    a.upload(d_mask, mask)
    before = a.download(d_goal, (3, n)).T.copy()
    a.sample_goals_dev(d_goal, d_mask)
    a.synchronize()
    after = a.download(d_goal, (3, n)).T
    expect = b.get_new_goal_joint_angles(mask)
    assert np.array_equal(after[mask == 1], expect[mask == 1]) and np.array_equal(after[mask == 0], before[mask == 0])
    a.close(); b.close()
    env = RoboyVecEnv(msj_robot, 512, seed=1, max_episode_length=5)
    env.reset()
    for _ in range(12):
        env.step(np.zeros((512, 8), np.float32))
    d_out = env.sim.malloc(64)
    env.stats_dev(d_out)
    env.sim.synchronize()
    host = env.stats()
    dev = env.sim.download(d_out, (8,), np.float64)
    assert np.array_equal(dev, np.array(list(host.values()))) and host["n_episodes"] >= 2 * 512
    env.close()


def _ball_joint_robot(msj_robot, n_tendons, seed):
    """A robot of the ball-joint class with another tendon count (random geometry, general inertia)."""
    from test_oracle import _random_ball_joint_robot
    desc = _random_ball_joint_robot(np.random.default_rng(seed), n_tendons)
    box = type(msj_robot._ACTION_SPACE)(low=-0.3, high=0.3, shape=(n_tendons,), dtype="float32")

    class Robot(type(msj_robot)):
        _DIM_ACTION = n_tendons
        _ACTION_SPACE = box

        @classmethod
        def get_description(cls):
            return desc
    return Robot()


@pytest.mark.parametrize("n_tendons", [1, 4, 6, 12, 16])
@pytest.mark.parametrize("integrator", ["euler", "rk4"])
def test_ball_joint_robots_with_other_tendon_counts_keep_the_closed_form(msj_robot, n_tendons, integrator):
    """Ball-joint robots with 1..16 tendons run the env-per-lane closed form with a run-time
    tendon count (not the 100x slower joint-tree kernel); both launch configurations."""
    from oracle.c_oracle import COracle
    robot = _ball_joint_robot(msj_robot, n_tendons, 40 + n_tendons)
    oracle = COracle(robot.get_description(), "f64")
    for n in (1, 1000, 66000):
        sim = _sim(robot, n, integrator=integrator)
        info = sim.info()
        assert info["kernel"] == KERNELS["env_per_lane"] and info["n_t"] == n_tendons
        assert info["bytes_per_env_step"] == 4 * (12 + n_tendons + 1)
        sim.close()
        _check_step(robot, oracle, n, integrator, 2, seed=n_tendons, kernel=KERNELS["env_per_lane"])
    sim = _sim(robot, 64)
    with pytest.raises(Exception, match="8 tendons"):
        sim.select_kernel(KERNELS["tendon_per_lane"])
    d = sim.malloc(4 * 64 * n_tendons)
    with pytest.raises(Exception, match="8-tendon"):
        sim.rollout_fused_dev(d, 1, 4, 1.0)
    sim.fill_actions_dev(d, 0)
    sim.rollout_dev(d, 1, 20, 0.3, use_graph=True)          # the graph path works for them as well
    sim.synchronize()
    assert np.isfinite(sim.read_state()[0]).all()
    sim.close()


def test_seventeen_tendons_fall_back_to_the_tree_kernel(msj_robot):
    from oracle.c_oracle import COracle
    robot = _ball_joint_robot(msj_robot, 17, 3)
    sim = _sim(robot, 10)
    assert sim.info()["kernel"] == 3          # RB_KERNEL_ENV_PER_WAVE
    sim.close()
    desc = robot.get_description()
    q, qd, sp = random_states(desc, 50, 2)
    sim = _sim(robot, 50)
    sim.set_state(q, qd)
    q1, qd1, _ = sim.forward_step_command(sp)
    qo, qdo, _ = COracle(desc, "f64").step(q.astype(np.float64), qd.astype(np.float64), sp.astype(np.float64), integrator=0)
    assert np.abs(q1 - qo).max() < 2e-5 and np.abs(qd1 - qdo).max() < 2e-5
    sim.close()


def test_sharding_is_invisible(msj_robot):
    """Two handles of 512 envs with env_id_offset 0 / 512 reproduce one handle
    of 1024 (random streams are keyed by the global env id)."""
    whole = _sim(msj_robot, 1024, seed=3)
    parts = [_sim(msj_robot, 512, seed=3, env_id_offset=o) for o in (0, 512)]
    def run(sim):
        d = sim.malloc(4 * sim.n_envs * 8)
        for t in range(20):
            sim.fill_actions_dev(d, t)
            sim.step_dev(d, 0.3)
        sim.synchronize()
        return sim.read_state(), sim.get_new_goal_joint_angles()
    (qw, qdw, fw), gw = run(whole)
    res = [run(p) for p in parts]
    assert np.array_equal(qw, np.concatenate([r[0][0] for r in res]))
    assert np.array_equal(qdw, np.concatenate([r[0][1] for r in res]))
    assert np.array_equal(gw, np.concatenate([r[1] for r in res]))
    for s in [whole] + parts:
        s.close()


@pytest.mark.parametrize("integrator", ["rk4", "euler"])
def test_shards_on_either_side_of_the_auto_thresholds(msj_robot, integrator):
    """ADVICE (round 4): RB_KERNEL_AUTO picks the kernel form by the HANDLE's batch size, so shards whose sizes straddle
    the thresholds (RK4 12 288 / 32 768 envs, Euler 4 096 / 16 384) run other forms than the whole batch: equal to a few ulp per
    step (the forms order their sums differently), and BIT-equal once every handle is pinned to one form - the contract written
    in include/roboy_sim.h at rb_select_kernel."""
    # eight lanes / two lanes / one lane per env; 65 536 envs in all: the env-per-lane form has two instances, one up to 65 536 envs
    # per handle (64-thread workgroups, tendons unrolled) and one above (256 threads, RK4 stages rolled: equal to ~1 ulp) - the
    # pinned comparison stays inside the first
    sizes = (8192, 20480, 36864) if integrator == "rk4" else (2048, 10240, 53248)
    total = sum(sizes)
    offsets = np.concatenate([[0], np.cumsum(sizes)[:-1]])

    def run(sim, steps=3):
        d = sim.malloc(4 * sim.n_envs * 8)
        for t in range(steps):
            sim.fill_actions_dev(d, t)
            sim.step_dev(d, 0.3)
        sim.synchronize()
        return sim.read_state()

    for pinned in (False, True):
        whole = _sim(msj_robot, total, seed=3, integrator=integrator)
        parts = [_sim(msj_robot, n, seed=3, env_id_offset=int(o), integrator=integrator) for n, o in zip(sizes, offsets)]
        if pinned:
            for s in [whole] + parts:
                s.select_kernel(KERNELS["env_per_lane"])
        else:
            assert whole.info()["kernel"] == KERNELS["env_per_lane"]
            assert [p.info()["kernel"] for p in parts] == [KERNELS["tendon_per_lane"], KERNELS["lane_pair"], KERNELS["env_per_lane"]]
        qw, qdw, fw = run(whole)
        res = [run(p) for p in parts]
        qp, qdp = np.concatenate([r[0] for r in res]), np.concatenate([r[1] for r in res])
        if pinned:
            assert np.array_equal(qw, qp) and np.array_equal(qdw, qdp)
        else:
            assert np.abs(qw - qp).max() < 3e-6 and np.abs(qdw - qdp).max() < 3e-5       # a few ulp of O(1) angles per step, three steps
            assert not (np.array_equal(qw, qp) and np.array_equal(qdw, qdp))              # (the forms do differ in the last bits)
        for s in [whole] + parts:
            s.close()


def test_pushing_into_the_boundary_becomes_and_stays_infeasible(msj_robot):
    """Restates test_simulation_client.py:54-68 for the batched client."""
    sim = _sim(msj_robot, 4)
    low = np.tile(msj_robot.get_action_space().low, (4, 1))
    feasible = np.ones(4, bool)
    for _ in range(1000):
        q, qd, feasible = sim.forward_step_command(low)
        if not feasible.any():
            break
    assert not feasible.any()
    q, qd, feasible = sim.forward_step_command(low)
    assert not feasible.any()
    assert msj_robot.get_joint_angles_space().contains(q[0])
    sim.close()


def test_unsupported_robot_is_refused_loudly(msj_robot):
    from gym_roboy_amd.envs.robots import RobotDescription, msj_platform_spec
    from gym_roboy_amd.envs.simulations import HipBatchSimulation
    from gym_roboy_amd._native import NativeError
    spec = msj_platform_spec()
    spec["joints"][1]["origin"] = [0.0, 0.0, 0.05]   # no longer a ball joint -> generic tree kernel
    spec["tendons"] = [dict(t, name="t%d" % i) for i in range(9) for t in spec["tendons"]][:65]   # > 64 tendons
    class Odd:
        @staticmethod
        def get_description():
            return RobotDescription(spec)
    with pytest.raises(NativeError, match="no HIP kernel"):
        HipBatchSimulation(Odd(), 4)


def test_random_streams_match_the_numpy_restatement_bit_for_bit(msj_robot):
    """Device Philox streams (synthetic actions, goals) vs oracle/philox_np.py,
    which is itself pinned to the Random123 known-answer vectors."""
    from oracle import philox_np as ph
    n, seed, off = 1000, 0x1234567890ABCDEF, (1 << 33) + 17
    sim = _sim(msj_robot, n, seed=seed, env_id_offset=off)
    ids = np.arange(off, off + n, dtype=np.uint64)
    d_act = sim.malloc(4 * n * 8)
    for step in (0, 1, 4000000000):
        sim.fill_actions_dev(d_act, step)
        sim.synchronize()
        got = sim.download(d_act, (n, 8))
        want = ph.actions(seed, ids, step, 8)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
        assert got.min() >= -1.0 and got.max() < 1.0
    desc = msj_robot.get_description()
    lo, hi = desc.q_lo.astype(np.float32), desc.q_hi.astype(np.float32)
    seen = []
    for draw in range(3):
        got = sim.get_new_goal_joint_angles()
        want = ph.goals(seed, ids, draw, lo, hi)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
        assert np.all(got >= lo) and np.all(got <= hi)
        seen.append(got)
    assert not np.allclose(seen[0], seen[1]) and not np.allclose(seen[1], seen[2])
    sim.close()


def test_states_outside_the_feasible_region_are_clamped_and_flagged(msj_robot):
    """An env handed a pose beyond a joint limit comes back clamped onto the
    limit with the outward velocity removed and is flagged infeasible; angles
    always satisfy RoboyRobot.new_state's assert (roboy_robot.py:76)."""
    desc = msj_robot.get_description()
    n = 64
    q = np.zeros((n, 3), np.float32); qd = np.zeros((n, 3), np.float32)
    q[:, 0] = 3.0; qd[:, 0] = 0.5          # far beyond +0.45, moving outward
    q[1::2, 1] = -3.0; qd[1::2, 1] = -0.5  # and beyond the lower limit for half of them
    for kernel in (1, 2):
        sim = _sim(msj_robot, n)
        sim.select_kernel(kernel)
        sim.set_state(q, qd)
        q1, qd1, f1 = sim.forward_step_command(np.zeros((n, 8), np.float32))
        assert not f1.any()
        assert np.all(q1[:, 0] == np.float32(desc.q_hi[0])) and np.all(qd1[:, 0] <= 0)
        assert np.all(q1[1::2, 1] == np.float32(desc.q_lo[1])) and np.all(qd1[1::2, 1] >= 0)
        assert msj_robot.get_joint_angles_space().contains(q1[0])
        sim.close()


def test_bad_arguments_raise(msj_robot):
    from gym_roboy_amd.envs.simulations import HipBatchSimulation
    with pytest.raises(ValueError):
        HipBatchSimulation(msj_robot, 0)
    with pytest.raises(ValueError):
        HipBatchSimulation(msj_robot, 8, integrator="verlet")
    sim = _sim(msj_robot, 8)
    with pytest.raises(ValueError):
        sim.forward_step_command(np.zeros((7, 8), np.float32))      # wrong batch
    with pytest.raises(ValueError):
        sim.set_state(np.zeros((8, 2), np.float32), np.zeros((8, 3), np.float32))
    with pytest.raises(ValueError):
        sim.step_dev(0)                                             # null device pointer
    with pytest.raises(ValueError):
        sim.select_kernel(9)
    sim.close()


def test_general_inertia_branch_matches_oracle(msj_robot):
    """Full inertia tensor, off-axis COM, tilted gravity: the non-"simple" branch
    of the rigid-body closed form, both kernel forms and both integrators."""
    from gym_roboy_amd.envs.robots import RobotDescription, msj_platform_spec
    from oracle.physics_np import TendonRobotOracle
    spec = msj_platform_spec()
    spec["joints"][2]["com"] = [0.012, -0.02, 0.06]
    spec["joints"][2]["inertia"] = [3.0e-4, 3.5e-4, 5.0e-4, 4.0e-5, -3.0e-5, 2.0e-5]
    spec["gravity"] = [1.0, -2.0, -9.0]
    desc = RobotDescription(spec)

    class Skewed(type(msj_robot)):
        @classmethod
        def get_description(cls):
            return desc
    oracle = TendonRobotOracle(desc)
    for kernel in (1, 2):
        for integrator in ("euler", "rk4"):
            _check_step(Skewed(), oracle, 1000, integrator, 1, seed=17, kernel=kernel)


@pytest.mark.parametrize("integrator", ["euler", "rk4"])
def test_long_rollout_with_extreme_actions_stays_finite_and_inside_the_boxes(msj_robot, integrator):
    """2 000 steps of i.i.d. full-range actions (5 episodes' worth): no NaN/inf,
    angles inside the feasible region, velocities inside the speed limit."""
    n = 4096
    desc = msj_robot.get_description()
    sim = _sim(msj_robot, n, integrator=integrator, seed=99)
    d_ring = sim.malloc(4 * 8 * n * 8)
    for r in range(8):
        sim.fill_actions_dev(d_ring + 4 * r * n * 8, r)
    sim.rollout_dev(d_ring, 8, 2000, 0.3, use_graph=True)
    sim.synchronize()
    q, qd, f = sim.read_state()
    assert np.isfinite(q).all() and np.isfinite(qd).all()
    assert np.all(q >= desc.q_lo.astype(np.float32)) and np.all(q <= desc.q_hi.astype(np.float32))
    assert np.all(np.abs(qd) <= np.float32(desc.qd_max) + 1e-7)
    assert q.std() > 0.01
    sim.close()


@pytest.mark.parametrize("integrator", ["euler", "rk4"])
def test_run_time_specialised_kernels_for_another_ball_joint_robot(msj_robot, integrator):
    """An 8-tendon ball-joint robot that is not MsjRobot (random geometry, general inertia): above 65 536 envs its
    env-per-lane kernels are compiled with hiprtc on the robot's own constants (rb_specialization = jit).  They must
    match the oracle, and the kernarg instances of the same source (ROBOY_SIM_JIT = 0) to rounding; MsjRobot itself
    reports the ahead-of-time table."""
    import os
    from oracle.c_oracle import COracle
    robot = _ball_joint_robot(msj_robot, 8, 77)
    desc = robot.get_description()
    n = 70001
    q, qd, sp = random_states(desc, n, 5)
    idx = np.arange(0, n, 97)
    qo, qdo, fo = COracle(desc, "f64").step(q[idx], qd[idx], sp[idx], integrator=0 if integrator == "euler" else 1)

    def run():
        sim = _sim(robot, n, integrator=integrator)
        spec = sim.specialization()
        sim.set_state(q, qd)
        out = sim.forward_step_command(sp)
        # the graph path must replay the module kernels as well
        d = sim.malloc(4 * n * 8)
        sim.fill_actions_dev(d, 0)
        sim.rollout_dev(d, 1, 12, 0.3, use_graph=True)
        sim.synchronize()
        assert np.isfinite(sim.read_state()[0]).all()
        sim.close()
        return spec, out
    spec, (q1, qd1, f1) = run()
    assert spec == "jit", spec
    assert np.abs(q1[idx] - qo).max() < 2e-5 and np.abs(qd1[idx] - qdo).max() < 2e-5
    os.environ["ROBOY_SIM_JIT"] = "0"
    try:
        spec0, (q0, qd0, f0) = run()
    finally:
        del os.environ["ROBOY_SIM_JIT"]
    assert spec0 == "kernarg"
    assert np.abs(q1 - q0).max() < 2e-6 and np.abs(qd1 - qd0).max() < 2e-6 and np.mean(f1 == f0) > 0.9999
    small = _sim(robot, 1000, integrator=integrator)
    assert small.specialization() == "kernarg"            # small batches keep the latency-oriented kernarg form
    small.close()
    msj = _sim(msj_robot, 70001, integrator=integrator)
    assert msj.specialization() == "table"
    msj.close()


def test_run_time_specialised_fused_env_kernel(msj_robot):
    """RoboyVecEnv over a non-MsjRobot 8-tendon ball-joint robot at a large batch: the fused env kernel comes out of
    the hiprtc module; same host replay as for MsjRobot (states / goals bit for bit against the plain step kernel,
    which runs out of the same module)."""
    import sys, os
    sys.path.insert(0, os.path.dirname(__file__))
    from host_env_model import HipStepper, HostEnvModel
    from gym_roboy_amd.envs.vec_env import RoboyVecEnv
    robot = _ball_joint_robot(msj_robot, 8, 78)
    n, seed, max_len = 66001, 4, 7
    vec = RoboyVecEnv(robot, n, seed=seed, max_episode_length=max_len, joint_vel_penalty=True)
    host = HostEnvModel(robot, HipStepper(robot, n, seed), n, seed, max_len, True, True, True)
    assert vec.sim.specialization() == "jit" and host.stepper.sim.specialization() == "jit"
    obs0 = vec.reset()
    host.goal = host.draw(np.ones(n, bool))
    assert np.array_equal(obs0[:, 6:], host.goal)
    rng = np.random.default_rng(9)
    for t in range(10):
        a = rng.uniform(-1, 1, (n, 8)).astype(np.float32)
        obs, rew, done, _ = vec.step(a)
        h_obs, h_rew, h_done, margin = host.step(a)
        same = done == h_done
        assert same.all() or (margin[~same] < 1e-5).all()
        assert np.array_equal(obs[same], h_obs[same].astype(np.float32))
        np.testing.assert_allclose(rew[same], h_rew[same], rtol=2e-5, atol=2e-4)
        assert same.all()
    vec.close(); host.stepper.close()


@pytest.mark.parametrize("kernel,n", [(1, 3000), (1, 70000), (5, 3000)])
def test_sub_ranges_on_two_streams_equal_the_whole_batch_step(msj_robot, kernel, n):
    """rb_step_range_dev: the batch stepped as two disjoint ranges on two streams (the second one ragged) against one launch
    over the whole batch - bit for bit; misaligned and out-of-range requests are refused, as are kernel forms that step whole
    batches only."""
    import torch
    desc = msj_robot.get_description()
    q, qd, sp = random_states(desc, n, 55)
    out = []
    for split in (False, True):
        sim = _sim(msj_robot, n, integrator="rk4")
        sim.select_kernel(kernel)
        sim.set_state(q, qd)
        act = torch.from_numpy(sp).cuda()
        if not split:
            sim.step_dev(act.data_ptr(), 1.0)
        else:
            assert sim.range_capable()
            mid = ((n // 2 + 255) // 256) * 256
            s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
            torch.cuda.synchronize()
            sim.step_range_dev(mid, n - mid, s2.cuda_stream, act.data_ptr(), 1.0)
            sim.step_range_dev(0, mid, s1.cuda_stream, act.data_ptr(), 1.0)
            torch.cuda.synchronize()
            with pytest.raises(Exception, match="multiple of 256"):
                sim.step_range_dev(100, 200, None, act.data_ptr(), 1.0)
            with pytest.raises(Exception, match="outside"):
                sim.step_range_dev(0, n + 1, None, act.data_ptr(), 1.0)
        sim.synchronize()
        out.append(sim.read_state())
        sim.close()
    for a, b in zip(*out):
        assert np.array_equal(a, b)
    sim = _sim(msj_robot, 1024)
    sim.select_kernel(KERNELS["tendon_per_lane"])
    assert not sim.range_capable()
    with pytest.raises(Exception, match="whole batches"):
        sim.step_range_dev(0, 512, None, torch.zeros(1024, 8, device="cuda").data_ptr(), 1.0)
    sim.close()


def test_destroy_drains_the_callers_stream_and_the_chain_streams(msj_robot):
    """rb_destroy / rb_select_kernel / rb_set_stream while a graph rollout in two chains is still in flight on a CALLER's
    stream: the handle drains every stream it uses before its graph executables, streams and buffers go (round-3 verdict:
    only the handle's own stream used to be synchronised)."""
    import torch
    n = 262144
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        for action in ("close", "select", "set_stream"):
            sim = _sim(msj_robot, n, integrator="rk4")
            sim.set_stream(st.cuda_stream)
            ring = torch.rand((4, n, 8), device="cuda") * 2 - 1
            assert sim.rollout_chains() == 2
            sim.rollout_dev(ring.data_ptr(), 4, 40, 0.3, use_graph=True)      # ~0.5 ms of work in flight on st and the chain stream
            if action == "select":
                sim.select_kernel(KERNELS["lane_pair"])                       # drops the cached graphs: must wait for them
                assert sim.info()["kernel"] == KERNELS["lane_pair"]
                sim.rollout_dev(ring.data_ptr(), 4, 16, 0.3, use_graph=True)
            elif action == "set_stream":
                sim.set_stream(None)                                          # back to the handle's own stream
                sim.rollout_dev(ring.data_ptr(), 4, 16, 0.3, use_graph=True)
            sim.close()                                                       # no synchronisation by the caller
    torch.cuda.synchronize()
    q = _sim(msj_robot, 16)
    assert np.all(q.read_state()[0] == 0)                                     # the device is alive
    q.close()


def test_close_waits_for_range_launches_on_the_callers_streams(msj_robot):
    """Round-4 verdict / ADVICE: rb_step_range_dev / rb_env_step_range_dev launch on streams the handle does not own; the
    handle remembers them (an event behind the last launch on each) and rb_destroy / rb_select_kernel wait for that work before
    buffers and graph state go - no join by the caller.  More distinct streams than the handle has entries (8): the oldest
    is waited for and reused.  The buffers are reusable afterwards and the device is alive."""
    import torch
    n = 262144
    act = torch.rand((n, 8), device="cuda") * 2 - 1
    for action in ("close", "select", "many_streams", "stream_gone"):
        sim = _sim(msj_robot, n, integrator="rk4")
        streams = [torch.cuda.Stream() for _ in range(12 if action == "many_streams" else 2)]
        torch.cuda.synchronize()
        mid = n // 2
        for rep in range(30):                       # ~0.3 ms of kernels in flight on streams the handle does not own
            for k, st in enumerate(streams):
                first = (k % 2) * mid
                sim.step_range_dev(first, mid, st.cuda_stream, act.data_ptr(), 0.3)
        if action == "select":
            sim.select_kernel(KERNELS["lane_pair"])                # waits for the range launches, then switches
            sim.step_range_dev(0, mid, streams[0].cuda_stream, act.data_ptr(), 0.3)
        if action == "stream_gone":
            del streams, st                                        # torch destroys the streams' handles lazily or not at all: either is fine
        sim.close()                                                # no synchronisation by the caller
    # fused env layer on a caller stream, closed at once
    from gym_roboy_amd.envs.vec_env import RoboyVecEnv
    env = RoboyVecEnv(msj_robot, n)
    st = torch.cuda.Stream()
    obs = torch.empty((n, 9), device="cuda"); rew = torch.empty(n, device="cuda"); done = torch.empty(n, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    for rep in range(30):
        env.step_range_dev(0, n // 2, st.cuda_stream, act.data_ptr(), obs.data_ptr(), rew.data_ptr(), done.data_ptr())
    env.close()
    torch.cuda.synchronize()
    assert bool(torch.isfinite(obs[: n // 2]).all())
    q = _sim(msj_robot, 16)
    assert np.all(q.read_state()[0] == 0)                          # the device is alive, memory reusable
    q.close()


def test_a_range_call_captured_on_a_callers_stream_defers_the_runtime_build():
    """ADVICE (round 4): the capture guard looks at the stream the range call launches on.  A robot whose env-per-lane kernels
    would be built at run time (hiprtc + module load: not capturable) sees its FIRST range call inside a capture of the
    caller's stream: the call is refused (no kernel form for sub-ranges yet) and the capture stays valid; after the capture
    the same call builds and runs."""
    import torch
    from random_robots import random_tree_robot
    robot, desc = random_tree_robot(11)
    n = 16384                                                      # AUTO wants the env-per-lane form from here on (hiprtc-built for this robot)
    sim = _sim(robot, n)
    act = torch.zeros((n, desc.n_t), device="cuda")
    st = torch.cuda.Stream()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(st):
        g.capture_begin()
        with pytest.raises(Exception, match="whole batches"):      # nothing built inside the capture: no range-capable form yet
            sim.step_range_dev(0, n // 2, st.cuda_stream, act.data_ptr(), 1.0)
        g.capture_end()                                            # raises if the capture was invalidated
    torch.cuda.synchronize()
    sim.step_range_dev(0, n // 2, st.cuda_stream, act.data_ptr(), 1.0)     # outside the capture: builds, then runs
    assert sim.range_capable() and sim.info()["kernel"] == 1 and sim.specialization() == "jit"
    sim.close()                                                    # waits for the launch on st
    assert np.isfinite(_sim(robot, 64).read_state()[0]).all()


def test_handles_give_their_device_memory_streams_and_graphs_back(msj_robot):
    """Production hygiene: create / use / destroy many handles - every robot class and kernel form, graph rollouts in two chains,
    the fused env layer, range launches on caller streams (events), statistics - and the free device memory returns to where it
    was (hipMemGetInfo through torch; torch's own caching allocator is kept out of it by allocating nothing in between)."""
    import torch
    from gym_roboy_amd.envs.robots import UpperBodyRobot
    from gym_roboy_amd.envs.vec_env import RoboyVecEnv
    upper = UpperBodyRobot()
    act_m = torch.zeros((131072, 8), device="cuda")
    act_u = torch.zeros((32768, 38), device="cuda")
    ring = torch.zeros((4, 131072, 8), device="cuda")
    obs_m = torch.zeros((131072, 9), device="cuda"); obs_u = torch.zeros((32768, 60), device="cuda")
    rew = torch.zeros(131072, device="cuda"); done = torch.zeros(131072, dtype=torch.int32, device="cuda")
    side = torch.cuda.Stream()

    def cycle():
        for n, kernel in ((512, 0), (20000, 0), (131072, 0), (131072, 5)):
            sim = _sim(msj_robot, n, integrator="rk4")
            if kernel:
                sim.select_kernel(kernel)
            sim.rollout_dev(ring.data_ptr(), 4, 24, 0.3, use_graph=True)
            if sim.range_capable():
                sim.step_range_dev(0, n // 2 // 256 * 256 or n, side.cuda_stream, act_m.data_ptr(), 0.3)
            sim.close()
        for n, kernel in ((300, 0), (300, 3), (20000, 0), (32768, 1)):
            sim = _sim(upper, n)
            if kernel:
                sim.select_kernel(kernel)
            sim.step_dev(act_u.data_ptr(), 0.3)
            sim.close()
        for robot, n, a, o in ((msj_robot, 20000, act_m, obs_m), (upper, 20000, act_u, obs_u)):
            env = RoboyVecEnv(robot, n)
            for _ in range(3):
                env.step_dev(a.data_ptr(), o.data_ptr(), rew.data_ptr(), done.data_ptr())
            env.stats()
            env.close()
        torch.cuda.synchronize()

    cycle()                                               # first use: code objects, the runtime's own pools
    free0 = torch.cuda.mem_get_info()[0]
    for _ in range(5):
        cycle()
    free1 = torch.cuda.mem_get_info()[0]
    assert free0 - free1 < 8 << 20, "device memory lost over 5 create / destroy cycles: %.1f MB" % ((free0 - free1) / 2 ** 20)


def test_a_refused_kernel_selection_changes_nothing():
    """ADVICE (round 3): rb_select_kernel validates before it writes - a refused request leaves rb_info, rb_specialization and the
    dispatch of the next step as they were (joint trees: the split form stays the library's choice)."""
    from gym_roboy_amd.envs.robots import UpperBodyRobot
    sim = _sim(UpperBodyRobot(), 4096)
    before = (sim.info()["kernel"], sim.specialization())
    assert before == (4, "table")
    for bad in (2, 5, 17):
        with pytest.raises(Exception):
            sim.select_kernel(bad)
        assert (sim.info()["kernel"], sim.specialization()) == before
    q0 = sim.forward_step_command(np.zeros((4096, 38), np.float32))[0]
    assert sim.info()["kernel"] == 4 and np.isfinite(q0).all()
    sim.close()
