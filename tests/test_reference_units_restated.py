"""The reference's own unit tests, restated against this package.

One test here per test in /root/reference/gym_roboy/envs/tests/
(test_robot_state.py, the non-integration half of test_roboy_env.py), same
assertions, same stub client; each cites the reference test it restates.  The
integration half (test_simulation_client.py and the `integration` params) is
restated against the HIP client in tests/test_simulation_client_gpu.py."""
from itertools import combinations

import numpy as np
import pytest

from gym_roboy_amd._gymcompat import spaces
from gym_roboy_amd.envs import RoboyEnv
from gym_roboy_amd.envs.roboy_env import _rescale_from_one_space_to_other
from gym_roboy_amd.envs.robots import MsjRobot, RobotState
from gym_roboy_amd.envs.simulations import StubSimulationClient

MSJ_ROBOT = MsjRobot()


@pytest.fixture
def sim_client():
    return StubSimulationClient(robot=MSJ_ROBOT)


@pytest.fixture
def roboy_env(sim_client):
    return RoboyEnv(simulation_client=sim_client)


# ---- test_robot_state.py -------------------------------------------------
def test_msj_robot_state_interpolate():                                   # :7-13
    a, b = MsjRobot.new_random_state(), MsjRobot.new_random_state()
    mid = RobotState.interpolate(a, b)
    assert np.allclose(mid.joint_angles, (a.joint_angles + b.joint_angles) / 2)
    assert np.allclose(mid.joint_vels, (a.joint_vels + b.joint_vels) / 2)


def test_msj_robot_state_new_random_zero_angle_state():                   # :16-19
    s = MsjRobot.new_random_zero_angles_state()
    assert np.allclose(s.joint_angles, 0) and not np.allclose(s.joint_vels, 0)


def test_msj_robot_state_new_random_zero_vel_state():                     # :22-25
    s = MsjRobot.new_random_zero_vels_state()
    assert np.allclose(s.joint_vels, 0) and not np.allclose(s.joint_angles, 0)


def test_robot_new_max_and_min_state():                                   # :28-37
    mx, mn = MSJ_ROBOT.new_max_state(), MSJ_ROBOT.new_min_state()
    assert np.allclose(mx.joint_angles, MSJ_ROBOT.get_joint_angles_space().high)
    assert np.allclose(mx.joint_vels, MSJ_ROBOT.get_joint_vels_space().high)
    assert np.allclose(mn.joint_angles, MSJ_ROBOT.get_joint_angles_space().low)
    assert np.allclose(mn.joint_vels, MSJ_ROBOT.get_joint_vels_space().low)


def test_robot_normalize_max_state():                                     # :40-50
    ones = np.ones(3)
    mx = MSJ_ROBOT.normalize_state(state=MSJ_ROBOT.new_max_state())
    mn = MSJ_ROBOT.normalize_state(state=MSJ_ROBOT.new_min_state())
    assert np.allclose(mx.joint_angles, ones) and np.allclose(mx.joint_vels, ones)
    assert np.allclose(mn.joint_angles, -ones) and np.allclose(mn.joint_vels, -ones)


def test_normalize_between_1_and_minus1():                                # :53-60
    high = np.random.random()
    low = high - np.abs(np.random.random())
    assert np.isclose(1, MSJ_ROBOT._normalize_between_minus1_and1(high, max_val=high, min_val=low))
    assert np.isclose(-1, MsjRobot._normalize_between_minus1_and1(low, max_val=high, min_val=low))


def test_robot_state_feasible_flag_must_be_bool():                        # roboy_robot.py:7-8 (typeguard)
    with pytest.raises(TypeError):
        RobotState([0, 0, 0], [0, 0, 0], is_feasible=1)
    with pytest.raises(AssertionError):
        MSJ_ROBOT.new_state(joint_angle=[4.0, 0, 0], joint_vel=[0, 0, 0], is_feasible=True)   # :76


# ---- test_roboy_env.py (unit-test-default params) ------------------------
def test_roboy_env_step(roboy_env):                                       # :28-33
    roboy_env.reset()
    obs, reward, done, _ = roboy_env.step(roboy_env.action_space.sample())
    assert isinstance(obs, np.ndarray) and isinstance(reward, float) and isinstance(done, bool)


def test_roboy_env_reset(roboy_env):                                      # :36-46
    nz = 2 * MSJ_ROBOT.get_joint_angles_space().shape[0]
    all_obs = [roboy_env.reset() for _ in range(5)]
    for obs in all_obs:
        assert isinstance(obs, np.ndarray) and np.allclose(obs[:nz], 0)
    for o1, o2 in combinations(all_obs, 2):
        assert np.allclose(o1[:nz], o2[:nz])


def test_roboy_env_new_goal_is_different_and_feasible(roboy_env):         # :49-57
    for _ in range(3):
        roboy_env._set_new_goal()
        old = roboy_env._goal_state
        roboy_env._set_new_goal()
        new = roboy_env._goal_state
        assert not np.allclose(old.joint_angles, new.joint_angles)
        assert np.all(MSJ_ROBOT.get_joint_angles_space().low <= new.joint_angles)
        assert np.all(new.joint_angles <= MSJ_ROBOT.get_joint_angles_space().high)


def test_roboy_env_reaching_goal_angle_delivers_maximum_reward(roboy_env):   # :60-68
    roboy_env.reset()
    roboy_env._set_new_goal(goal_joint_angle=roboy_env._last_state.joint_angles)
    _, reward, done, _ = roboy_env.step(np.zeros(len(roboy_env.action_space.low)))
    assert np.isclose(reward, roboy_env.reward_range[1])


def test_roboy_env_reaching_goal_joint_angle_but_moving_returns_done_equals_false(roboy_env):   # :71-79
    roboy_env.reset()
    roboy_env._set_new_goal(goal_joint_angle=roboy_env._last_state.joint_angles)
    roboy_env._last_state.joint_vels = MSJ_ROBOT.get_joint_vels_space().high
    assert not roboy_env._did_reach_goal(current_state=roboy_env._last_state, goal_state=roboy_env._goal_state)


def test_roboy_env_joint_vel_penalty_affects_worst_possible_reward(sim_client):   # :82-89
    env = RoboyEnv(simulation_client=sim_client, joint_vel_penalty=False)
    largest = np.linalg.norm(2 * np.ones(MSJ_ROBOT.get_joint_angles_space().shape))
    worst = -np.exp(largest) - abs(env._PENALTY_FOR_TOUCHING_BOUNDARY)
    assert np.isclose(env.reward_range[0], worst)
    env = RoboyEnv(simulation_client=sim_client, joint_vel_penalty=True)
    assert env.reward_range[0] < worst


def test_roboy_env_reward_is_lower_with_joint_vel_penalty(sim_client):    # :92-108
    goal = MsjRobot.new_random_state()
    goal.joint_vels = np.zeros_like(goal.joint_vels)
    sim_client.forward_step_command = lambda a: goal
    rewards = []
    action = None
    for pen in (False, True):
        env = RoboyEnv(simulation_client=sim_client, joint_vel_penalty=pen)
        env.reset()
        env._set_new_goal(goal_joint_angle=goal.joint_angles)
        action = env.action_space.sample() if action is None else action
        rewards.append(env.step(action=action)[1])
    assert rewards[0] > rewards[1]


def test_roboy_env_agent_gets_bonus_when_reaching_the_goal(sim_client):   # :111-122
    out = []
    for bonus in (False, True):
        env = RoboyEnv(simulation_client=sim_client, is_agent_getting_bonus_for_reaching_goal=bonus)
        env.reset()
        out.append(env.compute_reward(current_state=env._goal_state, goal_state=env._goal_state))
    assert np.allclose(out[1] - out[0], env._BONUS_FOR_REACHING_GOAL)


def test_roboy_env_render_does_nothing(roboy_env):                        # :125-126
    roboy_env.render()


@pytest.mark.parametrize("pen", [True, False], ids=["with joint_vel penalty", "no joint_vel penalty"])
def test_roboy_env_reward_monotonously_improves_during_approach(sim_client, pen):   # :129-167
    env = RoboyEnv(simulation_client=sim_client, joint_vel_penalty=pen)
    np.random.seed(0)
    env.seed(0)
    goal = MSJ_ROBOT.new_random_zero_vels_state()
    starts = [MSJ_ROBOT.new_random_state() for _ in range(40)]
    starts.append(MSJ_ROBOT.new_random_zero_vels_state())
    if pen:
        starts.append(MSJ_ROBOT.new_random_zero_angles_state())
    for cur in starts:
        seq = []
        for _ in range(7):
            seq.append(env.compute_reward(current_state=cur, goal_state=goal))
            cur = RobotState.interpolate(cur, goal)
        assert all(x < y for x, y in zip(seq, seq[1:]))


def test_roboy_env_maximum_episode_length(sim_client):                    # :170-180
    env = RoboyEnv(simulation_client=sim_client)
    env.reset()
    env.step_num = env._MAX_EPISODE_LENGTH - 1
    assert not env.step(np.zeros(env.action_space.shape))[2]
    assert not env._did_reach_goal(env._last_state, env._goal_state)
    assert env.step(env.action_space.sample())[2]


def test_roboy_reset_sets_step_number_to_one(roboy_env):                  # :183-188
    roboy_env.step(roboy_env.action_space.sample())
    assert roboy_env.step_num != 1
    roboy_env.reset()
    assert roboy_env.step_num == 1


def test_roboy_env_rescale_from_one_space_to_other(roboy_env):            # :195-213
    np.random.seed(0)
    dim = roboy_env.action_space.shape[0]
    def rand_space():
        return spaces.Box(low=-np.random.uniform(size=dim), high=np.random.uniform(size=dim), dtype="float32")
    sp_in, sp_out = rand_space(), rand_space()
    hi = _rescale_from_one_space_to_other(input_space=sp_in, output_space=sp_out, input_val=sp_in.high)
    lo = _rescale_from_one_space_to_other(input_space=sp_in, output_space=sp_out, input_val=sp_in.low)
    assert np.allclose(hi, sp_out.high) and np.allclose(lo, sp_out.low)


def test_step_rejects_actions_outside_the_unit_box(roboy_env):            # roboy_env.py:52
    roboy_env.reset()
    with pytest.raises(AssertionError):
        roboy_env.step(2.0 * np.ones(8, np.float32))
