"""The reference's integration tests (test_simulation_client.py:13-76 and the
`integration` params of test_roboy_env.py) against the ORACLE-backed client, on
CPU: the behavioural constraints the model spec was built to satisfy, checked
without a GPU.  (tests/test_simulation_client_gpu.py runs the same assertions
through the HIP client.)"""
from itertools import combinations

import numpy as np
import pytest

from gym_roboy_amd.envs import RoboyEnv
from gym_roboy_amd.envs.robots import MsjRobot, UpperBodyRobot
from oracle.cpu_simulation_client import CpuSimulationClient


@pytest.fixture(params=[MsjRobot, UpperBodyRobot], ids=["msj", "upper_body"])
def client(request):
    return CpuSimulationClient(robot=request.param())


def test_reset_gives_the_zero_state(client):                                        # :13-19
    s = client.forward_reset_command()
    assert np.allclose(0, s.joint_angles) and np.allclose(0, s.joint_vels) and s.is_feasible


def test_two_random_steps_change_the_state(client):                                 # :22-33
    rng = np.random.default_rng(0)
    box = client.robot.get_action_space()
    a = client.forward_step_command(rng.uniform(box.low, box.high).tolist())
    b = client.forward_step_command(rng.uniform(box.low, box.high).tolist())
    assert not np.allclose(a.joint_angles, b.joint_angles) and not np.allclose(a.joint_vels, b.joint_vels)


def test_read_state_does_not_change_the_state(client):                              # :36-44
    a, b = client.read_state(), client.read_state()
    assert np.array_equal(a.joint_angles, b.joint_angles) and np.array_equal(a.joint_vels, b.joint_vels)


def test_goals_are_different_and_inside_the_box(client):                            # :47-51
    goals = [client.get_new_goal_joint_angles() for _ in range(5)]
    for g1, g2 in combinations(goals, 2):
        assert not np.allclose(g1, g2)
    assert all(client.robot.get_joint_angles_space().contains(g.astype(np.float32)) for g in goals)


def test_pushing_on_the_boundary_goes_infeasible_and_does_not_reset(client):        # :54-68
    client.forward_reset_command()
    strong = client.robot.get_action_space().low.tolist()
    client.forward_reset_command = lambda: pytest.fail("should not call this")
    for _ in range(1000):
        state = client.forward_step_command(action=strong)
        if not state.is_feasible:
            break
    assert not state.is_feasible
    assert not client.forward_step_command(action=strong).is_feasible


def test_roboy_env_over_the_physics_client():
    """test_roboy_env.py integration params: reset obs, max reward at the goal
    with zero action (:60-68), episode length (:170-180)."""
    env = RoboyEnv(simulation_client=CpuSimulationClient(robot=MsjRobot()))
    obs = env.reset()
    assert np.allclose(obs[:6], 0) and obs.dtype == np.float64
    env._set_new_goal(goal_joint_angle=env._last_state.joint_angles)
    _, reward, done, _ = env.step(np.zeros(8))
    assert np.isclose(reward, env.reward_range[1]) and done
    env.reset()
    obs, reward, done, _ = env.step(env.action_space.sample())
    assert isinstance(reward, float) and isinstance(done, bool) and env.observation_space.shape == obs.shape
    env.step_num = env._MAX_EPISODE_LENGTH
    assert env.step(env.action_space.sample())[2]
