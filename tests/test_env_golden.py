"""Env layer vs golden vectors captured from the REFERENCE's Python
(tests/golden/env_layer.json, written by tests/golden/make_env_golden.py from
/root/reference with explicit inputs).  Covers SURVEY.md §8c's list: rescale,
normalize_state, compute_reward (4 flag combinations x feasible/infeasible x
at-goal/off-goal), _did_reach_goal either side of both thresholds,
reward_range, observation layout and dtype, episode length, interpolate.

The reference evaluates these in float64 on float64 states, and the rescale in
float32; this package uses the same numpy operations, so values are compared
to 1e-12 relative (bit-exact for the float32 rescale)."""
import contextlib
import io
import json
import os

import numpy as np
import pytest

from gym_roboy_amd.envs import RoboyEnv
from gym_roboy_amd.envs.roboy_env import _l2_distance, _rescale_from_one_space_to_other
from gym_roboy_amd.envs.robots import MsjRobot, RobotState
from gym_roboy_amd.envs.simulations import MockSimulationClient, SimulationClient
from gym_roboy_amd._gymcompat import spaces

FX = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "env_layer.json")))
ROBOT = MsjRobot()
RTOL = 1e-12


def _env(pen=False, bonus=True, client=None):
    return RoboyEnv(simulation_client=client or MockSimulationClient(robot=ROBOT), joint_vel_penalty=pen,
                    is_agent_getting_bonus_for_reaching_goal=bonus)


def test_spaces_match_reference():
    s = FX["spaces"]
    for box, lo, hi in ((ROBOT.get_joint_angles_space(), "angle_low", "angle_high"),
                        (ROBOT.get_joint_vels_space(), "vel_low", "vel_high"),
                        (ROBOT.get_action_space(), "action_low", "action_high")):
        assert box.low.dtype == np.float32
        assert np.array_equal(box.low.astype(np.float64), s[lo])
        assert np.array_equal(box.high.astype(np.float64), s[hi])


def test_env_constants_match_reference():
    c, env = FX["env_constants"], _env()
    assert float(env._MAX_DISTANCE_JOINT_ANGLE) == c["max_distance_joint_angle"]
    assert float(env._MAX_DISTANCE_JOINT_VELS) == c["max_distance_joint_vels"]
    assert env._PENALTY_FOR_TOUCHING_BOUNDARY == c["penalty"]
    assert env._BONUS_FOR_REACHING_GOAL == c["bonus"]
    assert env._MAX_EPISODE_LENGTH == c["max_episode_length"]
    assert env.step_num == c["initial_step_num"]
    assert np.array_equal(env._GOAL_JOINT_VEL, c["goal_joint_vel"])
    assert env.observation_space.low.dtype == np.dtype(c["obs_dtype"])
    assert np.array_equal(env.observation_space.low.astype(np.float64), c["obs_low"])
    assert np.array_equal(env.observation_space.high.astype(np.float64), c["obs_high"])
    assert np.array_equal(env.action_space.low.astype(np.float64), c["action_space_low"])
    assert env.action_space.shape == (8,) and env.observation_space.shape == (9,)


@pytest.mark.parametrize("row", FX["reward_range"], ids=lambda r: "pen%d_bonus%d" % (r["joint_vel_penalty"], r["bonus"]))
def test_reward_range_matches_reference(row):
    env = _env(row["joint_vel_penalty"], row["bonus"])
    assert isinstance(env.reward_range[0], float)
    np.testing.assert_allclose(env.reward_range, row["range"], rtol=1e-7)   # float32 arithmetic inside


def test_rescale_is_bit_exact():
    env = _env()
    for case in FX["rescale"]:
        x = np.array(case["input"], dtype=np.float32)
        y = _rescale_from_one_space_to_other(input_val=x, input_space=env.action_space,
                                             output_space=ROBOT.get_action_space())
        assert str(y.dtype) == case["output_dtype"]
        assert np.array_equal(y.astype(np.float64), case["output"])
    g = FX["rescale_general"]
    sp_in = spaces.Box(low=np.array(g["in_low"]), high=np.array(g["in_high"]), dtype="float32")
    sp_out = spaces.Box(low=np.array(g["out_low"]), high=np.array(g["out_high"]), dtype="float32")
    y = _rescale_from_one_space_to_other(input_val=np.array(g["input"], dtype=np.float32),
                                         input_space=sp_in, output_space=sp_out)
    assert np.array_equal(y.astype(np.float64), g["output"])


def test_normalize_and_interpolate_match_reference():
    for row in FX["normalize"]:
        st = ROBOT.normalize_state(RobotState(np.array(row["q"]), np.array(row["qd"]), True))
        np.testing.assert_allclose(st.joint_angles, row["q_norm"], rtol=RTOL)
        np.testing.assert_allclose(st.joint_vels, row["qd_norm"], rtol=RTOL)
    c = FX["normalize_corners"]
    mx = ROBOT.normalize_state(ROBOT.new_max_state())
    mn = ROBOT.normalize_state(ROBOT.new_min_state())
    assert np.array_equal(mx.joint_angles, c["max_q"]) and np.array_equal(mx.joint_vels, c["max_qd"])
    assert np.array_equal(mn.joint_angles, c["min_q"]) and np.array_equal(mn.joint_vels, c["min_qd"])
    assert mx.is_feasible == c["max_feasible"]
    i = FX["interpolate"]
    mid = RobotState.interpolate(RobotState(np.array(i["a"][0]), np.array(i["a"][1]), i["a"][2]),
                                 RobotState(np.array(i["b"][0]), np.array(i["b"][1]), i["b"][2]))
    assert np.array_equal(mid.joint_angles, i["mid"][0]) and np.array_equal(mid.joint_vels, i["mid"][1])
    assert mid.is_feasible == i["mid"][2]
    assert _l2_distance(np.array([np.inf, 1.0]), np.array([np.inf, 0.0])) == FX["l2_distance"]["inf_minus_inf"]


def test_reward_and_goal_detection_match_reference():
    envs = {(p, b): _env(p, b) for p in (False, True) for b in (False, True)}
    e0 = envs[(False, True)]
    assert float(e0._MAX_DISTANCE_JOINT_ANGLE / 200) == FX["goal_thresholds"]["angle"]
    assert float(e0._MAX_DISTANCE_JOINT_VELS / 5) == FX["goal_thresholds"]["vel"]
    n_reached = 0
    for row in FX["reward_cases"]:
        cur = RobotState(np.array(row["q"]), np.array(row["qd"]), row["feasible"])
        goal = ROBOT.new_state(joint_angle=np.array(row["goal_q"]), joint_vel=e0._GOAL_JOINT_VEL, is_feasible=True)
        with contextlib.redirect_stdout(io.StringIO()):
            assert e0._did_reach_goal(current_state=cur, goal_state=goal) == row["reached"]
            for (p, b), env in envs.items():
                r = env.compute_reward(current_state=cur, goal_state=goal)
                assert isinstance(r, float)
                np.testing.assert_allclose(r, row["reward"]["pen%d_bonus%d" % (p, b)], rtol=RTOL)
        n_reached += row["reached"]
    assert 0 < n_reached < len(FX["reward_cases"])


class ScriptedClient(SimulationClient):
    def __init__(self, states, goals):
        self.robot = ROBOT
        self.states, self.goals = list(states), list(goals)
        self.i = self.g = 0
        self.received = []

    def read_state(self):
        return ROBOT.new_state(joint_angle=[0.0] * 3, joint_vel=[0.0] * 3, is_feasible=True)

    def forward_step_command(self, action):
        self.received.append(list(action))
        q, qd, ok = self.states[self.i]
        self.i += 1
        return ROBOT.new_state(joint_angle=list(q), joint_vel=list(qd), is_feasible=bool(ok))

    def forward_reset_command(self):
        return self.read_state()

    def get_new_goal_joint_angles(self):
        g = self.goals[self.g % len(self.goals)]
        self.g += 1
        return np.array(g)


def test_scripted_episode_matches_reference_step_by_step():
    ep = FX["episode"]
    client = ScriptedClient(ep["script"], ep["goals"])
    env = _env(client=client)
    with contextlib.redirect_stdout(io.StringIO()):
        obs0 = env.reset()
        assert np.array_equal(obs0, ep["reset_obs"])
        for t, want in enumerate(ep["steps"]):
            obs, rew, done, info = env.step(np.array(ep["actions"][t], dtype=np.float32))
            assert str(obs.dtype) == want["obs_dtype"] == "float64"
            assert np.array_equal(obs, want["obs"])
            assert type(rew).__name__ == ep["reward_type"] and type(done).__name__ == ep["done_type"]
            np.testing.assert_allclose(rew, want["reward"], rtol=RTOL)
            assert done == want["done"] and env.step_num == want["step_num"]
            assert np.array_equal(env._goal_state.joint_angles, want["goal_after"])
            assert info == {}
    assert client.received == ep["received_setpoints"]   # python floats of the float32 rescale
    assert any(s["done"] for s in ep["steps"]) and not all(s["done"] for s in ep["steps"])


def test_episode_length_matches_reference():
    el = FX["episode_length"]
    env = _env(client=ScriptedClient([([0.1, 0.1, 0.1], [0.0] * 3, True)] * 4, FX["episode"]["goals"]))
    env.reset()
    env.step_num = env._MAX_EPISODE_LENGTH - 1
    zero = np.zeros(8, np.float32)
    dones = [env.step(zero)[2] for _ in range(3)]
    assert dones == [el["done_at_399_plus_1"], el["done_at_400_plus_1"], el["done_after"]]
    assert env.step_num == el["step_num_after"]


def test_registry_ids():
    import gym_roboy_amd
    for env_id in FX["registry_ids"] + ["msj-control-v1"]:
        env = gym_roboy_amd.make(env_id, simulation_client=MockSimulationClient(robot=ROBOT))
        assert isinstance(env, RoboyEnv)
