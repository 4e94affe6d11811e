"""The reference's integration tests (gym_roboy/envs/tests/test_simulation_client.py
and the `integration` params of test_roboy_env.py, which need a live CARDSflow)
restated against ``HipSimulationClient``: same assertions, in-process GPU physics."""
from itertools import combinations

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def client():
    from gym_roboy_amd.envs.robots import MsjRobot
    from gym_roboy_amd.envs.simulations import HipSimulationClient
    c = HipSimulationClient(robot=MsjRobot())
    yield c
    c.close()


def _random_setpoints(robot, rng):
    box = robot.get_action_space()
    return rng.uniform(box.low, box.high).astype(np.float32).tolist()


def test_simulation_client_reset(client):                                  # :13-19
    s = client.forward_reset_command()
    assert np.allclose([0, 0, 0], s.joint_angles) and np.allclose([0, 0, 0], s.joint_vels)
    assert s.is_feasible is True and s.joint_angles.dtype == np.float64


def test_simulation_client_step(client):                                   # :22-33
    rng = np.random.default_rng(0)
    a = client.forward_step_command(_random_setpoints(client.robot, rng))
    b = client.forward_step_command(_random_setpoints(client.robot, rng))
    assert not np.allclose(a.joint_angles, b.joint_angles)
    assert not np.allclose(a.joint_vels, b.joint_vels)


def test_simulation_client_read_state(client):                             # :36-44
    a, b = client.read_state(), client.read_state()
    assert np.allclose(a.joint_angles, b.joint_angles) and np.allclose(a.joint_vels, b.joint_vels)


def test_simulation_client_get_new_goal_joint_angles_results_are_different(client):   # :47-51
    goals = [client.get_new_goal_joint_angles() for _ in range(5)]
    for g1, g2 in combinations(goals, 2):
        assert not np.allclose(g1, g2)
    for g in goals:
        assert client.robot.get_joint_angles_space().contains(g)


def test_simulation_client_stepping_on_the_boundary_does_not_reset(client):   # :54-68
    client.forward_reset_command()
    strong = client.robot.get_action_space().low.tolist()
    original = client.forward_reset_command
    client.forward_reset_command = lambda: pytest.fail("should not call this")
    try:
        for _ in range(1000):
            state = client.forward_step_command(action=strong)
            if not state.is_feasible:
                break
        assert not state.is_feasible
        assert not client.forward_step_command(action=strong).is_feasible
    finally:
        client.forward_reset_command = original


def test_simulation_client_load_test():                                    # :71-76
    from gym_roboy_amd.envs.robots import MsjRobot
    from gym_roboy_amd.envs.simulations import HipSimulationClient
    c = HipSimulationClient(robot=MsjRobot())
    for _ in range(100):
        c.forward_reset_command()
        c.get_new_goal_joint_angles()
    c.close()


def test_wrong_action_length_is_a_type_error(client):                      # typeguard List[float], ros_..py:48-49
    with pytest.raises(TypeError):
        client.forward_step_command([0.0] * 7)


def test_roboy_env_over_the_hip_client_behaves_like_the_reference_env(client):
    """integration params of test_roboy_env.py: step types (:28-33), reset obs
    (:36-46), goals (:49-57), max reward at the goal with zero action (:60-68)."""
    import gym_roboy_amd
    from gym_roboy_amd.envs import RoboyEnv
    env = RoboyEnv(simulation_client=client)
    obs = env.reset()
    assert np.allclose(obs[:6], 0) and obs.dtype == np.float64
    obs, reward, done, info = env.step(env.action_space.sample())
    assert isinstance(obs, np.ndarray) and isinstance(reward, float) and isinstance(done, bool)
    env._set_new_goal(); g1 = env._goal_state.joint_angles
    env._set_new_goal(); g2 = env._goal_state.joint_angles
    assert not np.allclose(g1, g2)
    env.reset()
    env._set_new_goal(goal_joint_angle=env._last_state.joint_angles)
    _, reward, done, _ = env.step(np.zeros(8))
    assert np.isclose(reward, env.reward_range[1]) and done
    # gym.make with no kwargs builds the HIP client (the drop-in default)
    made = gym_roboy_amd.make("msj-control-v1")
    assert np.allclose(made.reset()[:6], 0)
