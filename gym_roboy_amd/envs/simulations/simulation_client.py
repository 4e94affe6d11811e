"""The simulator plugin interface and the physics-free stub.

Same names and semantics as
``/root/reference/gym_roboy/envs/simulations/simulation_client.py:6-47``:
``RoboyEnv`` talks to the robot only through ``robot`` and these four methods.
``MockSimulationClient`` is the name BASELINE.json uses for the stub (the
reference's tests bind it to ``MOCK_SIM_CLIENT``, ``test_roboy_env.py:12``).
"""
import numpy as np

from ..robots import RobotState, RoboyRobot


class SimulationClient:
    """How ``RoboyEnv`` drives a robot.  Implementations: the stub below (no
    physics) and ``HipSimulationClient`` (MI355X kernels)."""
    robot = RoboyRobot()

    def read_state(self) -> RobotState:
        raise NotImplementedError

    def forward_step_command(self, action) -> RobotState:
        raise NotImplementedError

    def forward_reset_command(self) -> RobotState:
        raise NotImplementedError

    def get_new_goal_joint_angles(self) -> np.ndarray:
        raise NotImplementedError


class StubSimulationClient(SimulationClient):
    """Unit-test double: a zero action keeps the state, any other action jumps
    to a fresh random state; reset gives the zero state; goals are random."""

    def __init__(self, robot: RoboyRobot):
        self.robot = robot
        self._state = robot.new_random_state()

    def read_state(self) -> RobotState:
        return self._state

    def forward_step_command(self, action) -> RobotState:
        assert len(action) == self.robot.get_action_space().shape[0]
        if np.allclose(action, 0):
            return self._state
        return self.robot.new_random_state()

    def forward_reset_command(self) -> RobotState:
        self._state = self.robot.new_zero_state()
        return self._state

    def get_new_goal_joint_angles(self):
        return self.robot.new_random_state().joint_angles


MockSimulationClient = StubSimulationClient
