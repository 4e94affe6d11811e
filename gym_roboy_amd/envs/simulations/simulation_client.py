"""The simulator plugin interface and the physics-free stub.

Same names and semantics as
``/root/reference/gym_roboy/envs/simulations/simulation_client.py:6-47``:
``RoboyEnv`` talks to the robot only through ``robot`` and these four methods.
``MockSimulationClient`` is the name BASELINE.json uses for the stub (the
reference's tests bind it to ``MOCK_SIM_CLIENT``, ``test_roboy_env.py:12``).
"""
import numpy as np

from ..robots import RobotState, RoboyRobot


def _not_implemented(method: str):
    raise NotImplementedError(
        "SimulationClient.%s: use HipSimulationClient (MI355X physics) or the stub" % method)


class SimulationClient:
    """How ``RoboyEnv`` drives a robot.

    Implementations in this package: ``StubSimulationClient`` below (no physics,
    CPU, for plumbing tests) and ``HipSimulationClient`` (in-process MI355X
    kernels).  The reference's only real implementation goes through ROS to an
    external simulator (``ros_simulation_client.py``).
    """
    robot = RoboyRobot()

    def read_state(self) -> RobotState:
        """Current joint angles / velocities / feasibility; must not change them."""
        _not_implemented("read_state")

    def forward_step_command(self, action) -> RobotState:
        """Hold the tendon set-points ``action`` for one step; return the new state."""
        _not_implemented("forward_step_command")

    def forward_reset_command(self) -> RobotState:
        """Put the robot back into the zero pose at rest; return that state."""
        _not_implemented("forward_reset_command")

    def get_new_goal_joint_angles(self) -> np.ndarray:
        """A random feasible joint-angle vector to use as the next goal."""
        _not_implemented("get_new_goal_joint_angles")


class StubSimulationClient(SimulationClient):
    """Unit-test double with the reference stub's behaviour: an all-zero action
    leaves the state untouched, any other action teleports to a fresh random
    state (which is *not* remembered), reset gives the zero state, goals are
    random angle vectors."""

    def __init__(self, robot: RoboyRobot):
        self.robot = robot                       # the attribute RoboyEnv reads (roboy_env.py:19)
        self._n_tendons = robot.get_action_space().shape[0]
        self._current = robot.new_random_state()

    def read_state(self):
        return self._current

    def forward_step_command(self, action):
        assert len(action) == self._n_tendons, "expected %d set-points" % self._n_tendons
        idle = bool(np.allclose(action, 0))
        return self._current if idle else self.robot.new_random_state()

    def forward_reset_command(self):
        self._current = self.robot.new_zero_state()
        return self._current

    def get_new_goal_joint_angles(self):
        return self.robot.new_random_state().joint_angles


MockSimulationClient = StubSimulationClient
