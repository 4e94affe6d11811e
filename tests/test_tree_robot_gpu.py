"""GPU parity of the joint-tree kernels against the fp64 oracle, on the synthetic 20-DOF / 38-tendon upper body
(BASELINE.json configs[3]) and on MSJ variants that are not ball joints, in every kernel form: env-per-lane
(rb_kernel 1: code generated for the robot - compiled ahead of time for the upper body, by hiprtc otherwise), its
split form (rb_kernel 4: several waves per group of 64 envs, one per set of branches - the library's choice for the
upper body up to 16 384 envs) and octets (rb_kernel 3: two envs per wave, eight lanes per link, tables in LDS).
Tolerance 2e-5 on the state after one env step."""
import numpy as np
import pytest

from conftest import random_states

pytestmark = pytest.mark.gpu
TOL = 2e-5


@pytest.fixture(scope="module")
def upper_body():
    from gym_roboy_amd.envs.robots import UpperBodyRobot
    return UpperBodyRobot()


@pytest.fixture(scope="module")
def upper_oracle(upper_body):
    from oracle.c_oracle import COracle
    return COracle(upper_body.get_description(), "f64")


LANE, OCTET, SPLIT, SPLIT2 = 1, 3, 4, 6      # rb_kernel: RB_KERNEL_ENV_PER_LANE, _ENV_PER_WAVE, _ENV_PER_LANE_SPLIT, _ENV_PER_LANE_SPLIT2 (lean two-part form, round 5)


def _check(robot, oracle, n, integrator, nsub, seed, kernel=None, expect=OCTET):
    """kernel None: the library's own choice, which must be `expect`."""
    from gym_roboy_amd.envs.simulations import HipBatchSimulation
    desc = robot.get_description()
    q, qd, sp = random_states(desc, n, seed)
    sim = HipBatchSimulation(robot, n, integrator=integrator, n_substeps=nsub)
    if kernel is not None:
        sim.select_kernel(kernel)
        expect = kernel
    assert sim.info()["kernel"] == expect
    sim.set_state(q, qd)
    q1, qd1, f1 = sim.forward_step_command(sp)
    qo, qdo, fo = oracle.step(q, qd, sp, integrator=0 if integrator == "euler" else 1, n_substeps=nsub)
    assert np.abs(q1 - qo).max() < TOL, np.abs(q1 - qo).max()
    assert np.abs(qd1 - qdo).max() < TOL, np.abs(qd1 - qdo).max()
    near = np.minimum(np.abs(qo - desc.q_lo), np.abs(qo - desc.q_hi)).min(axis=1) < 1e-5
    assert not np.any((f1 != fo) & ~near)
    sim.close()


@pytest.mark.parametrize("integrator", ["euler", "rk4"])
@pytest.mark.parametrize("nsub", [1, 2])
@pytest.mark.parametrize("n", [1, 257])
@pytest.mark.parametrize("kernel", [None, LANE, OCTET, SPLIT2])
def test_upper_body_step_matches_oracle(upper_body, upper_oracle, integrator, nsub, n, kernel):
    # the library's choice for the committed upper body at these batch sizes is the split form compiled ahead of time
    _check(upper_body, upper_oracle, n, integrator, nsub, seed=n + nsub, kernel=kernel, expect=SPLIT)


def test_kernel_forms_of_the_upper_body_agree_with_each_other(upper_body):
    """Same states and set-points through all three forms: the differences are rounding (other summation orders at the
    trunk, other instruction streams), far inside the tolerance against the oracle."""
    from gym_roboy_amd.envs.simulations import HipBatchSimulation
    desc = upper_body.get_description()
    n = 300
    q, qd, sp = random_states(desc, n, 21)
    out = {}
    for kernel in (LANE, SPLIT, OCTET, SPLIT2):
        for integ in ("euler", "rk4"):
            sim = HipBatchSimulation(upper_body, n, integrator=integ)
            sim.select_kernel(kernel)
            sim.set_state(q, qd)
            out[kernel, integ] = sim.forward_step_command(sp)
            sim.close()
    for integ in ("euler", "rk4"):
        for kernel in (SPLIT, OCTET, SPLIT2):
            # (three instruction streams with three summation orders - the lane form writes the two arms as one stream of
            # pair values: half the tolerance against the oracle)
            assert np.abs(out[kernel, integ][0] - out[LANE, integ][0]).max() < 1e-5
            assert np.abs(out[kernel, integ][1] - out[LANE, integ][1]).max() < 1e-5


def test_ragged_last_wave_and_unaligned_rows_in_the_one_wave_kernels(upper_body):
    """Edge cases of the row movement of the one-wave kernels (a wave's rows are one contiguous run behind a range-checked buffer
    resource): a ragged last wave (61 live envs) gives what the same envs give inside full waves, bit for bit; so do action /
    observation arrays at a 4-byte offset in the fused env layer (legal there for joint trees).  (Written for a 16-bytes-per-lane
    form of the row movement - LDS-DMA loads, dwordx4 stores - that was measured and not kept: profiles/r5_a/x4_ab.log.)"""
    import torch
    from gym_roboy_amd.envs.simulations import HipBatchSimulation
    from gym_roboy_amd.envs.vec_env import RoboyVecEnv
    desc = upper_body.get_description()
    full, ragged = 64 * 5, 64 * 5 - 3
    q, qd, sp = random_states(desc, full, 17)
    out = {}
    for n in (full, ragged):
        sim = HipBatchSimulation(upper_body, n)
        sim.select_kernel(LANE)
        sim.set_state(q[:n], qd[:n])
        out[n] = sim.forward_step_command(sp[:n])
        sim.close()
    for a, b in zip(out[full], out[ragged]):
        assert np.array_equal(a[:ragged], b)
    obs = []
    for n, shift in ((full, 0), (full, 1), (ragged, 0), (ragged, 1)):
        vec = RoboyVecEnv(upper_body, n, seed=2)
        vec.sim.select_kernel(LANE)
        vec.reset()
        vec.sim.set_state(q[:n], qd[:n])
        abuf = torch.zeros(n * desc.n_t + 4, dtype=torch.float32, device="cuda")
        act = abuf[shift:shift + n * desc.n_t]
        act.copy_(torch.from_numpy(np.clip(sp[:n] / np.float32(0.3), -1, 1).reshape(-1)))
        assert (act.data_ptr() % 16 == 0) == (shift == 0)
        obuf = torch.zeros(n * 3 * desc.n_q + 4, dtype=torch.float32, device="cuda")
        o = obuf[shift:shift + n * 3 * desc.n_q]
        rew = torch.zeros(n, device="cuda"); done = torch.zeros(n, dtype=torch.int32, device="cuda")
        vec.step_dev(act.data_ptr(), o.data_ptr(), rew.data_ptr(), done.data_ptr())
        torch.cuda.synchronize()
        obs.append((o.cpu().numpy().reshape(n, -1)[:ragged, :2 * desc.n_q].copy(), rew.cpu().numpy()[:ragged].copy()))
        vec.close()
    for other in obs[1:]:
        assert np.array_equal(obs[0][0], other[0]) and np.array_equal(obs[0][1], other[1])
    assert np.abs(obs[0][0][:, :desc.n_q] - out[full][0][:ragged]).max() < 1e-6


def test_upper_body_specialization_is_the_ahead_of_time_table(upper_body):
    from gym_roboy_amd.envs.simulations import HipBatchSimulation
    sim = HipBatchSimulation(upper_body, 64)
    assert sim.specialization() == "table" and sim.info()["kernel"] == SPLIT
    sim.select_kernel(OCTET)
    assert sim.specialization() == "kernarg" and sim.info()["kernel"] == OCTET
    sim.select_kernel(LANE)
    assert sim.specialization() == "table" and sim.info()["kernel"] == LANE
    sim.select_kernel(0)
    assert sim.info()["kernel"] == SPLIT
    sim.close()
    mid = HipBatchSimulation(upper_body, 16384 + 64)          # beyond one five-wave workgroup per CU: two part waves per env group, two workgroups per CU
    assert mid.info()["kernel"] == SPLIT2 and mid.specialization() == "table" and not mid.range_capable()
    mid.close()
    big = HipBatchSimulation(upper_body, 32768 + 64)          # beyond that: one wave per 64 envs (a wave on every SIMD from 65 536 envs on)
    assert big.info()["kernel"] == LANE and big.specialization() == "table" and big.range_capable()
    big.close()


def test_upper_body_numpy_oracle_agrees_too(upper_body):
    """The C oracle used above against the numpy statement of the spec, on this robot."""
    from oracle.c_oracle import COracle
    from oracle.physics_np import TendonRobotOracle
    desc = upper_body.get_description()
    q, qd, sp = random_states(desc, 16, 3)
    a = TendonRobotOracle(desc).step(q.astype(np.float64), qd.astype(np.float64), sp.astype(np.float64), integrator=1)
    b = COracle(desc, "f64").step(q, qd, sp, integrator=1)
    assert np.abs(a[0] - b[0]).max() < 1e-12 and np.abs(a[1] - b[1]).max() < 1e-11


def test_upper_body_rest_pose_is_an_equilibrium_and_rollout_tracks_oracle(upper_body, upper_oracle):
    from gym_roboy_amd.envs.simulations import HipBatchSimulation
    n = 64
    sim = HipBatchSimulation(upper_body, n)
    q, qd, f = sim.forward_step_command(np.zeros((n, 38), np.float32))
    # fp32 gravity terms cancel to ~1e-8 rad, not to an exact zero as in the closed form
    assert np.abs(q).max() < 1e-6 and np.abs(qd).max() < 1e-5 and f.all()
    sim.forward_reset_command()
    rng = np.random.default_rng(2)
    qo = np.zeros((n, 20)); qdo = np.zeros((n, 20))
    worst = 0.0
    for t in range(120):
        if t % 20 == 0:
            sp = rng.uniform(-0.3, 0.3, (n, 38)).astype(np.float32)
        q, qd, f = sim.forward_step_command(sp)
        qo, qdo, fo = upper_oracle.step(qo, qdo, sp)
        worst = max(worst, np.abs(q - qo).max(), np.abs(qd - qdo).max())
    assert worst < 1e-3, worst
    assert np.abs(q).max() > 0.05
    sim.close()


def test_upper_body_boundary_behaviour_and_goals(upper_body):
    """test_simulation_client.py:47-68 for the second robot."""
    from gym_roboy_amd.envs.simulations import HipBatchSimulation
    sim = HipBatchSimulation(upper_body, 2)
    low = np.tile(upper_body.get_action_space().low, (2, 1))
    for _ in range(1000):
        q, qd, f = sim.forward_step_command(low)
        if not f.any():
            break
    assert not f.any()
    assert not sim.forward_step_command(low)[2].any()
    desc = upper_body.get_description()
    g1, g2 = sim.get_new_goal_joint_angles(), sim.get_new_goal_joint_angles()
    assert g1.shape == (2, 20) and not np.allclose(g1, g2)
    assert np.all(g1 >= desc.q_lo.astype(np.float32)) and np.all(g1 <= desc.q_hi.astype(np.float32))
    sim.close()


def test_msj_variant_outside_the_ball_joint_class_takes_the_tree_kernel(msj_robot):
    """Same tendons, joint 1 moved 5 cm up: not a ball joint any more, so the
    closed form refuses and the generic kernel must take over and still match."""
    from gym_roboy_amd.envs.robots import RobotDescription, msj_platform_spec
    from oracle.c_oracle import COracle
    spec = msj_platform_spec()
    spec["joints"][1]["origin"] = [0.0, 0.0, 0.05]
    desc = RobotDescription(spec)

    class OffsetMsj(type(msj_robot)):
        @classmethod
        def get_description(cls):
            return desc
    for integrator in ("euler", "rk4"):
        _check(OffsetMsj(), COracle(desc, "f64"), 100, integrator, 1, seed=4)
    # and the env-per-lane form, built by hiprtc for this robot (a batch this small takes the octets on its own)
    for integrator in ("euler", "rk4"):
        _check(OffsetMsj(), COracle(desc, "f64"), 100, integrator, 2, seed=5, kernel=LANE)


def test_tree_kernel_on_the_msj_robot_equals_the_closed_form(msj_robot, msj_oracle):
    """A description that IS ball-joint class but carries an (inert) 4th joint is
    routed to the tree kernel; its first three joints must follow the MSJ oracle."""
    from gym_roboy_amd.envs.robots import RobotDescription, msj_platform_spec
    from oracle.c_oracle import COracle
    spec = msj_platform_spec()
    spec["joints"].append({"name": "idle", "parent": 2, "axis": [1, 0, 0], "origin": [0, 0, 0.1], "mass": 0.01,
                           "com": [0, 0, 0.0], "inertia": [1e-6] * 3 + [0, 0, 0], "armature": 0.2, "damping": 0.8,
                           "limit": [-0.4, 0.4], "max_velocity": 0.5})
    desc = RobotDescription(spec)

    class Msj4(type(msj_robot)):
        @classmethod
        def get_description(cls):
            return desc
    _check(Msj4(), COracle(desc, "f64"), 65, "euler", 1, seed=8)


@pytest.mark.parametrize("kernel", [None, LANE, OCTET, SPLIT2])      # None: the library's choice for 130 envs, the split form
@pytest.mark.parametrize("auto_reset,integrator", [(True, "euler"), (False, "euler"), (True, "rk4")])
def test_fused_env_layer_on_the_upper_body_matches_host_replay(upper_body, auto_reset, integrator, kernel):
    """RoboyVecEnv over the joint-tree kernel: same replay check as for MsjRobot
    (tests/test_env_layer_gpu.py): states and goals bit for bit against the plain
    tree kernel + numpy Philox, reward/done against reward.py in float64."""
    import sys, os
    sys.path.insert(0, os.path.dirname(__file__))
    from host_env_model import HipStepper, HostEnvModel
    from gym_roboy_amd.envs.vec_env import RoboyVecEnv
    n, seed, max_len = 130, 3, 9
    vec = RoboyVecEnv(upper_body, n, seed=seed, auto_reset=auto_reset, max_episode_length=max_len, joint_vel_penalty=True,
                      integrator=integrator)
    stepper = HipStepper(upper_body, n, seed, integrator=integrator)
    if kernel is not None:                                # the fused kernel and the plain step it is replayed with, in the same form
        vec.sim.select_kernel(kernel); stepper.sim.select_kernel(kernel)
    assert vec.sim.info()["kernel"] == (kernel or SPLIT) == stepper.sim.info()["kernel"]
    host = HostEnvModel(upper_body, stepper, n, seed, max_len, True, True, auto_reset)
    obs0 = vec.reset()
    host.goal = host.draw(np.ones(n, bool))
    assert obs0.shape == (n, 60) and not obs0[:, :40].any() and np.array_equal(obs0[:, 40:], host.goal)
    rng = np.random.default_rng(1)
    n_done = 0
    for t in range(25):
        a = rng.uniform(-1, 1, (n, 38)).astype(np.float32)
        obs, rew, done, _ = vec.step(a)
        h_obs, h_rew, h_done, margin = host.step(a)
        assert np.array_equal(done, h_done) or (margin[done != h_done] < 1e-5).all()
        same = done == h_done
        assert same.all()
        assert np.array_equal(obs, h_obs.astype(np.float32))
        np.testing.assert_allclose(rew, h_rew, rtol=3e-5, atol=3e-4)
        n_done += int(done.sum())
    assert n_done >= 2 * n
    st = vec.stats()
    assert st["n_env_steps"] == 25 * n and st["n_episodes"] == n_done
    got = np.array([st[k] for k in ("sum_return", "sum_return_sq", "n_episodes", "sum_length", "n_goal_reached",
                                    "n_infeasible_steps", "n_env_steps", "sum_reward")])
    np.testing.assert_allclose(got, host.stats, rtol=1e-4, atol=5e-2)
    vec.close(); host.stepper.close()


def _hand_robot(msj_robot, n_fingers=6):
    """A palm on one revolute with `n_fingers` two-joint fingers: the tree level of the finger bases is wider
    than one pass of a wave holds (2 envs x 8 octet slots x 8 lanes > 64), so the joint-tree kernel takes its
    multi-pass form (no register chaining), and a link with more than 4 children in other octets gathers the
    rest through the child list.  Tendons: two per finger, each crossing two links; one routed palm -> base -> palm
    -> tip with a same-link segment in between."""
    import math
    from gym_roboy_amd.envs.robots import RobotDescription
    from gym_roboy_amd.envs.robots.description import FORMAT_TAG

    def joint(name, parent, axis, origin, mass, com, limit):
        return {"name": name, "parent": parent, "axis": axis, "origin": origin, "mass": mass, "com": com,
                "inertia": [2e-4 * mass / 0.05, 2e-4 * mass / 0.05, 1e-4 * mass / 0.05, 0.0, 0.0, 0.0] if mass else [0.0] * 6,
                "armature": 0.05, "damping": 0.3, "limit": [-limit, limit], "max_velocity": math.pi / 6}
    joints = [joint("palm", -1, [0, 1, 0], [0, 0, 0.0], 0.4, [0, 0, 0.04], 0.5)]
    tendons = []
    for f in range(n_fingers):
        ang = 2 * math.pi * f / n_fingers
        ox, oy = 0.05 * math.cos(ang), 0.05 * math.sin(ang)
        b = len(joints)
        joints.append(joint("f%d_base" % f, 0, [-math.sin(ang), math.cos(ang), 0.0], [ox, oy, 0.08], 0.05, [0, 0, 0.02], 0.6))
        joints.append(joint("f%d_tip" % f, b, [-math.sin(ang), math.cos(ang), 0.0], [0, 0, 0.04], 0.03, [0, 0, 0.015], 0.6))
        for sgn in (1.0, -1.0):
            px, py = ox + sgn * 0.012 * math.cos(ang), oy + sgn * 0.012 * math.sin(ang)
            tendons.append({"name": "f%d_%s" % (f, "flex" if sgn > 0 else "ext"), "f_max": 20.0,
                            "via_points": [{"link": -1, "pos": [px * 1.2, py * 1.2, -0.03]},
                                           {"link": 0, "pos": [px, py, 0.05]}, {"link": 0, "pos": [px, py, 0.075]},
                                           {"link": b, "pos": [sgn * 0.012 * math.cos(ang), sgn * 0.012 * math.sin(ang), 0.03]},
                                           {"link": b + 1, "pos": [sgn * 0.010 * math.cos(ang), sgn * 0.010 * math.sin(ang), 0.03]}]})
    spec = {"format": FORMAT_TAG, "name": "hand", "gravity": [0.0, 0.0, -9.81], "joints": joints, "tendons": tendons,
            "muscle": {"kp": 10.0, "setpoint_scale": 0.02, "v_max": 8.0, "fl_width": 0.45, "kpe": 4.0, "e0": 0.6,
                       "fv_a": 0.25, "fv_n": 1.5}}
    desc = RobotDescription(spec)

    class Hand(type(msj_robot)):
        @classmethod
        def get_description(cls):
            return desc
    return Hand(), desc


@pytest.mark.parametrize("integrator", ["euler", "rk4"])
def test_wide_tree_takes_the_multi_pass_form_and_matches_the_oracle(msj_robot, integrator):
    from oracle.c_oracle import COracle
    robot, desc = _hand_robot(msj_robot)
    assert desc.n_q == 13 and desc.n_t == 12
    _check(robot, COracle(desc, "f64"), 301, integrator, 1, seed=11)
    _check(robot, COracle(desc, "f64"), 7, integrator, 2, seed=12)
