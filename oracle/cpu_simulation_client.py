"""TEST INFRASTRUCTURE - a ``SimulationClient`` backed by the C oracle.

Lets the CPU test suite run the reference's integration tests
(``/root/reference/gym_roboy/envs/tests/test_simulation_client.py``) against the
model without a GPU, and gives bench.py the "reference architecture" baseline
(one Python ``RoboyEnv`` per env, ``train_parallel.py:19-29``).  It lives under
oracle/ because it must never be a product path: the product's clients are
HIP-only.  Goals come from the same Philox stream as the device (philox_np).
"""
import numpy as np

from gym_roboy_amd.envs.robots import RobotState, RoboyRobot
from gym_roboy_amd.envs.simulations import SimulationClient

from . import philox_np as ph
from .c_oracle import COracle


class CpuSimulationClient(SimulationClient):

    def __init__(self, robot: RoboyRobot, process_idx: int = 1, integrator=0, n_substeps: int = 1, seed: int = 0):
        self.robot = robot
        self._desc = robot.get_description()
        self._orc = COracle(self._desc, "f32")           # the precision the device steps in
        self._integrator, self._nsub, self._seed, self._env_id = integrator, n_substeps, seed, int(process_idx)
        n_q = self._desc.n_q
        self._q = np.zeros((1, n_q), np.float32)
        self._qd = np.zeros((1, n_q), np.float32)
        self._f = np.ones(1, np.uint8)
        self._draw = 0

    def _state(self) -> RobotState:
        return self.robot.new_state(joint_angle=self._q[0].astype(np.float64),
                                    joint_vel=self._qd[0].astype(np.float64), is_feasible=bool(self._f[0]))

    def read_state(self) -> RobotState:
        return self._state()

    def forward_step_command(self, action) -> RobotState:
        sp = np.ascontiguousarray(np.asarray(action, dtype=np.float32)[None, :])
        if sp.shape != (1, self._desc.n_t):
            raise TypeError("action must be a sequence of %d floats" % self._desc.n_t)
        self._orc.step_inplace(self._q, self._qd, sp, self._f, integrator=self._integrator, n_substeps=self._nsub)
        return self._state()

    def forward_reset_command(self) -> RobotState:
        self._q[:] = 0; self._qd[:] = 0; self._f[:] = 1
        return self._state()

    def get_new_goal_joint_angles(self) -> np.ndarray:
        g = ph.goals(self._seed, np.array([self._env_id], dtype=np.uint64), self._draw,
                     self._desc.q_lo.astype(np.float32), self._desc.q_hi.astype(np.float32))
        self._draw += 1
        return g[0].astype(np.float64)
