"""The reference's own env-layer vectors (tests/golden/env_layer.json, captured from
/root/reference/gym_roboy/envs/roboy_env.py:51-134 by tests/golden/make_env_golden.py)
fed through the fused GPU kernel ``rb_env_step_dev``.

How a recorded (state, goal, flags) triple reaches the kernel: the kernel always takes a
physics step, so the robot is "parked" - actuators so heavy (armature 1e9, no damping, no
gravity, negligible muscle force) that one step leaves the velocity untouched and moves
the angles by exactly h * qd.  Starting from q - h*qd the step therefore lands on the
recorded state (to 1 ulp of fp32), and the kernel's observation / reward / done for it
are compared with what the reference returned.  Infeasible rows use the physics' own
mechanism: a joint limit placed at the recorded angle, approached from outside, so the
step clamps onto it (bit-exact) and flags the env infeasible
(ros_simulation_client.py:40-46 -> roboy_env.py:102-103).

Tolerances: the kernel evaluates the reward in fp32 (reference: float64) - rtol 2e-5 /
atol 2e-4 as in tests/test_env_layer_gpu.py; done flags are compared where the recorded
state is further than 1e-5 from either threshold.
"""
import json
import os

import numpy as np
import pytest

from gym_roboy_amd.envs.robots import MsjRobot, RobotDescription, msj_platform_spec

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "env_layer.json")
H = 0.1   # ros_simulation_client.py:22
# Every kernel form the fused env step of an 8-tendon ball-joint robot can take, BY NAME (rb_select_kernel + an assertion on the row
# the launch takes): 1 = one env per lane (msj_env_step_kernel - what every batch above 32 768 envs runs), 2 = eight lanes per env
# (msj_env_step_tendon_per_lane), 5 = two lanes per env (msj_env_step_mirror_pairs).  Never "whatever the library picks at this
# size": round 5's thresholds moved under such a test and the env-per-lane kernel lost the reference's vectors.
# tests/test_dispatch_table.py fails if the dispatch table holds a form this list lacks.
ENV_FORMS = (1, 2, 5)
FORM_NAMES = {1: "env_per_lane", 2: "tendon_per_lane", 5: "lane_pair"}


def pin_form(vec, form):
    """Select the form and assert the fused env step's next launch takes a row of that form."""
    vec.sim.select_kernel(form)
    row = vec.sim.dispatch("env_step")
    assert row["kernel"] == form and "/env_step/%s/" % FORM_NAMES[form] in row["id"], row
    return row


def parked_robot(limits=None):
    """MsjRobot boxes (what the env layer reads) over a description that does not
    accelerate: per-joint limits default to wider than the +-pi angle box."""
    spec = msj_platform_spec()
    spec["gravity"] = [0.0, 0.0, 0.0]
    for k, j in enumerate(spec["joints"]):
        j["armature"] = 1.0e9
        j["damping"] = 0.0
        j["max_velocity"] = 10.0
        j["limit"] = [-3.3, 3.3] if limits is None else list(limits[k])
    for t in spec["tendons"]:
        t["f_max"] = 1.0e-6
    desc = RobotDescription(spec)

    class ParkedMsjRobot(MsjRobot):
        @classmethod
        def get_description(cls):
            return desc
    return ParkedMsjRobot()


def pre_state(q, qd):
    """State from which one parked step lands on (q, qd)."""
    qd32 = np.asarray(qd, np.float32)
    return (np.asarray(q, np.float64) - H * qd32.astype(np.float64)).astype(np.float32), qd32


def limits_hitting(q, qd):
    """Joint limits and a start state such that the parked step clamps one joint exactly onto
    its recorded angle without touching its velocity, and flags the env infeasible.  limit()
    clamps to the bound and drops only the OUTWARD velocity component, and a description's
    limits must bracket the zero pose, so the joint j used is one whose recorded angle and
    velocity have opposite signs (upper bound = q[j] > 0 approached from above with v <= 0, or
    lower bound = q[j] < 0 approached from below with v >= 0).  Returns None if no joint qualifies."""
    q32 = np.asarray(q, np.float32).astype(np.float64)
    q_pre, qd32 = pre_state(q, qd)
    for j in range(3):
        lim = [[-3.3, 3.3] for _ in range(3)]
        if q32[j] > 1e-3 and qd32[j] <= 0:
            lim[j][1] = float(q32[j])
            q_pre[j] = np.float32(q32[j] + 0.05 + H * abs(float(qd32[j])))
            return lim, q_pre, qd32, j
        if q32[j] < -1e-3 and qd32[j] >= 0:
            lim[j][0] = float(q32[j])
            q_pre[j] = np.float32(q32[j] - 0.05 - H * abs(float(qd32[j])))
            return lim, q_pre, qd32, j
    return None


def test_parked_robot_lands_on_the_recorded_state_cpu():
    """The construction itself, checked on the CPU oracle (fp32 build: the device's precision)."""
    from oracle.c_oracle import COracle
    rng = np.random.default_rng(1)
    q = rng.uniform(-3, 3, (32, 3)); qd = rng.uniform(-0.5, 0.5, (32, 3))
    orc = COracle(parked_robot().get_description(), "f32")
    q_pre, qd32 = pre_state(q, qd)
    q1, qd1, feas = orc.step(q_pre, qd32, np.zeros((32, 8), np.float32))
    assert feas.all() and np.abs(qd1 - qd32).max() < 1e-9
    assert np.abs(q1 - q.astype(np.float32)).max() < 5e-7
    hit = 0
    for i in range(16):
        case = limits_hitting(q[i], qd[i])
        if case is None:
            continue
        lim, qp, v, j = case
        o = COracle(parked_robot(lim).get_description(), "f32")
        qa, va, fa = o.step(qp[None], v[None], np.zeros((1, 8), np.float32))
        assert not fa[0] and qa[0, j] == np.float32(q[i, j]) and np.abs(va[0] - v).max() < 1e-9
        assert np.abs(qa[0] - q[i].astype(np.float32)).max() < 5e-7
        hit += 1
    assert hit >= 8


def _fixture():
    with open(GOLDEN) as fh:
        return json.load(fh)


def _margin(fx, q, qd, goal):
    da = np.linalg.norm(np.asarray(q, np.float64) - goal, axis=-1)
    dv = np.linalg.norm(np.asarray(qd, np.float64), axis=-1)
    return np.minimum(np.abs(da - fx["goal_thresholds"]["angle"]), np.abs(dv - fx["goal_thresholds"]["vel"]))


# (form, copies of the 40 recorded rows): every form by name on the 40 rows; the env-per-lane form also in its large-batch
# configuration - the rows tiled to 66 560 envs (256-thread workgroups; this robot's constants are not the ahead-of-time table's, so
# the library compiles ITS instance with hiprtc: msj_env_step_kernel<., 256, ., Const8, true>, the headline env kernel's family) and
# once with hiprtc off (the kernarg instance of the same configuration); "auto": nothing selected, 40 960 envs - above every threshold
# of the other two forms, so the library's own choice must be the env-per-lane kernel
GOLDEN_CASES = [(1, 1, None), (2, 1, None), (5, 1, None), (1, 1664, "1"), (1, 1664, "0"), (0, 1024, None)]


@pytest.mark.gpu
@pytest.mark.parametrize("form,copies,jit", GOLDEN_CASES, ids=["env_per_lane", "tendon_per_lane", "lane_pair", "env_per_lane-66560-hiprtc",
                                                              "env_per_lane-66560-kernarg", "auto-40960"])
@pytest.mark.parametrize("pen,bonus", [(False, False), (False, True), (True, False), (True, True)])
def test_reward_cases_through_the_fused_kernel(pen, bonus, form, copies, jit, monkeypatch):
    from gym_roboy_amd.envs.vec_env import RoboyVecEnv
    if jit is not None:
        monkeypatch.setenv("ROBOY_SIM_JIT", jit)
    fx = _fixture()
    key = "pen%d_bonus%d" % (pen, bonus)
    feasible = [c for c in fx["reward_cases"] if c["feasible"]]
    assert len(feasible) == 40
    tile = lambda a: np.tile(np.asarray(a), (copies,) + (1,) * (np.asarray(a).ndim - 1))
    q = tile([c["q"] for c in feasible]); qd = tile([c["qd"] for c in feasible])
    goal = tile([c["goal_q"] for c in feasible])
    want_r = tile([c["reward"][key] for c in feasible])
    want_reached = tile([c["reached"] for c in feasible])
    n = 40 * copies
    vec = RoboyVecEnv(parked_robot(), n, seed=3, joint_vel_penalty=pen,
                      is_agent_getting_bonus_for_reaching_goal=bonus, auto_reset=False)
    if form:
        row = pin_form(vec, form)
    else:
        row = vec.sim.dispatch("env_step")
        assert row["kernel"] == 1, row                  # above 32 768 envs the library's own choice is one env per lane
    if copies > 1:
        assert row["block"] == (256 if n > 65536 else 64) and row["constants"] == {None: 0, "0": 0, "1": 2}[jit], row
    vec.reset()
    q_pre, qd32 = pre_state(q, qd)
    vec.sim.set_state(q_pre, qd32)
    vec.set_goal(goal, step_num=np.full(n, 7, np.uint32))
    obs, rew, done, _ = vec.step(np.zeros((n, 8), np.float32))
    # observation = [q, qd, goal in effect during the step] (roboy_env.py:62,75-80)
    assert np.abs(obs[:, 0:3] - q.astype(np.float32)).max() < 5e-7
    assert np.abs(obs[:, 3:6] - qd32).max() < 1e-9          # 1e9 kg m^2 of armature: |dv| ~ 1e-17
    assert np.array_equal(obs[:, 6:9], goal.astype(np.float32))
    np.testing.assert_allclose(rew, want_r, rtol=2e-5, atol=2e-4)
    clear = _margin(fx, q, qd, goal) > 1e-5
    assert clear.sum() >= 30 * copies and want_reached[clear].any() and (~want_reached[clear]).any()
    assert np.array_equal(done[clear], want_reached[clear])        # step counter far from the limit: done == reached
    # the goal is resampled exactly where done was returned (roboy_env.py:67-68)
    obs2, _, _, _ = vec.step(np.zeros((n, 8), np.float32))
    changed = np.any(obs2[:, 6:9] != obs[:, 6:9], axis=1)
    assert np.array_equal(changed, done)
    assert vec.sim.dispatch("env_step")["id"] == row["id"]
    vec.close()


@pytest.mark.gpu
@pytest.mark.parametrize("form", ENV_FORMS)
def test_infeasible_reward_cases_through_the_fused_kernel(form):
    """Rows recorded with is_feasible = False: the joint limit sits at the recorded angle,
    the step clamps onto it and the kernel subtracts the boundary penalty (roboy_env.py:102-103).  (A limit on one joint only
    breaks the mirror symmetry the two-lanes-per-env form needs: that form is checked on the rows whose robot keeps it.)"""
    from gym_roboy_amd.envs.vec_env import RoboyVecEnv
    fx = _fixture()
    rows = [c for c in fx["reward_cases"] if not c["feasible"]]
    assert len(rows) == 40
    checked_done = covered = refused = 0
    for c in rows:
        case = limits_hitting(c["q"], c["qd"])
        if case is None:
            continue
        covered += 1
        lim, q_pre, qd32, j = case
        robot = parked_robot(lim)
        goal = np.array([c["goal_q"]])
        for pen in (False, True):
            for bonus in (False, True):
                vec = RoboyVecEnv(robot, 1, seed=1, joint_vel_penalty=pen,
                                  is_agent_getting_bonus_for_reaching_goal=bonus, auto_reset=False)
                try:
                    pin_form(vec, form)
                except Exception:
                    assert form == 5                     # (no mirror plane with this limit: the library refuses the pair form)
                    refused += 1
                    vec.close()
                    continue
                vec.reset()
                vec.sim.set_state(q_pre[None], qd32[None])
                vec.set_goal(goal, step_num=np.array([3], np.uint32))
                obs, rew, done, _ = vec.step(np.zeros((1, 8), np.float32))
                _, _, feas = vec.sim.read_state()
                assert not feas[0]
                assert obs[0, j] == np.float32(c["q"][j])                     # clamped onto the limit
                assert np.abs(obs[0, 0:3] - np.asarray(c["q"], np.float32)).max() < 5e-7
                assert np.abs(obs[0, 3:6] - qd32).max() < 1e-9
                np.testing.assert_allclose(rew[0], c["reward"]["pen%d_bonus%d" % (pen, bonus)], rtol=2e-5, atol=2e-4)
                if _margin(fx, c["q"], c["qd"], goal[0]) > 1e-5:
                    assert bool(done[0]) == c["reached"]
                    checked_done += 1
                vec.close()
    assert covered >= 30
    if form == 5:
        assert refused + checked_done > 0
    else:
        assert refused == 0 and checked_done >= 80


@pytest.mark.gpu
@pytest.mark.parametrize("form", ENV_FORMS)
def test_scripted_episode_through_the_fused_kernel(form):
    """The 12-step episode recorded from the reference (default flags: no velocity penalty,
    bonus on): every step's (state, goal, step counter) replayed as one env of a batch."""
    from gym_roboy_amd.envs.vec_env import RoboyVecEnv
    fx = _fixture()
    ep = fx["episode"]
    steps, script = ep["steps"], ep["script"]
    feas_idx = [t for t in range(len(steps)) if script[t][2]]
    infeas_idx = [t for t in range(len(steps)) if not script[t][2]]
    assert len(infeas_idx) >= 2 and any(steps[t]["done"] for t in feas_idx)

    def check(vec, rows, obs, rew, done):
        for k, t in enumerate(rows):
            want = np.asarray(steps[t]["obs"])
            assert np.abs(obs[k] - want.astype(np.float32)).max() < 5e-7
            np.testing.assert_allclose(rew[k], steps[t]["reward"], rtol=2e-5, atol=2e-4)
            assert bool(done[k]) == steps[t]["done"]

    n = len(feas_idx)
    vec = RoboyVecEnv(parked_robot(), n, seed=2, auto_reset=False)
    pin_form(vec, form)
    vec.reset()
    q = np.array([script[t][0] for t in feas_idx]); qd = np.array([script[t][1] for t in feas_idx])
    goal = np.array([steps[t]["obs"][6:9] for t in feas_idx])
    q_pre, qd32 = pre_state(q, qd)
    vec.sim.set_state(q_pre, qd32)
    vec.set_goal(goal, step_num=np.array([steps[t]["step_num"] - 1 for t in feas_idx], np.uint32))
    obs, rew, done, _ = vec.step(np.asarray([ep["actions"][t] for t in feas_idx], np.float32))
    check(vec, feas_idx, obs, rew, done)
    obs2, _, _, _ = vec.step(np.zeros((n, 8), np.float32))
    goal_changed = np.any(obs2[:, 6:9] != obs[:, 6:9], axis=1)
    recorded_change = np.array([steps[t]["goal_after"] != steps[t]["obs"][6:9] for t in feas_idx])
    assert np.array_equal(goal_changed, recorded_change)
    vec.close()
    replayed = 0
    for t in infeas_idx:
        case = limits_hitting(script[t][0], script[t][1])
        if case is None:
            continue
        replayed += 1
        lim, q_pre, qd32, _ = case
        vec = RoboyVecEnv(parked_robot(lim), 1, seed=2, auto_reset=False)
        pin_form(vec, form if form != 5 else 1)        # (a one-sided limit has no mirror plane: the env-per-lane form stands in)
        vec.reset()
        vec.sim.set_state(q_pre[None], qd32[None])
        vec.set_goal(np.array([steps[t]["obs"][6:9]]), step_num=np.array([steps[t]["step_num"] - 1], np.uint32))
        obs, rew, done, _ = vec.step(np.asarray([ep["actions"][t]], np.float32))
        check(vec, [t], obs, rew, done)
        vec.close()
    assert replayed >= 1


@pytest.mark.gpu
@pytest.mark.parametrize("form", ENV_FORMS)
def test_episode_length_through_the_fused_kernel(form):
    """done when step_num > 400 (roboy_env.py:72-73; fixture 'episode_length' recorded from the reference)."""
    from gym_roboy_amd.envs.vec_env import RoboyVecEnv
    fx = _fixture()["episode_length"]
    vec = RoboyVecEnv(parked_robot(), 3, seed=4, auto_reset=False)
    pin_form(vec, form)
    vec.reset()
    far = np.full((3, 3), 1.5, np.float32)
    vec.sim.set_state(np.full((3, 3), 0.1, np.float32), np.zeros((3, 3), np.float32))
    vec.set_goal(far, step_num=np.array([398, 399, 400], np.uint32))
    _, _, done, _ = vec.step(np.zeros((3, 8), np.float32))
    assert list(done) == [False, fx["done_at_399_plus_1"], fx["done_at_400_plus_1"]]
    vec.close()
