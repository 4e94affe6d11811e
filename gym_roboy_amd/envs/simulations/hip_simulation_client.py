"""In-process MI355X simulation clients (replace the ROS -> CARDSflow round-trip).

``HipBatchSimulation`` owns N environments in HBM and is the throughput path:
``forward_step_command(actions[N, n_t])`` is one kernel launch.
``HipSimulationClient`` is the single-env drop-in for the reference's
``RosSimulationClient`` (``gym_roboy/envs/simulations/ros_simulation_client.py:12-81``):
same constructor convention ``Client(robot, process_idx=1, ...)``, same four
methods, same return types.  Both call ``libroboy_sim.so`` through ctypes and
raise if it is missing or no GPU is visible; there is no CPU fallback.
"""
import ctypes

import numpy as np

from ... import _native as nat
from ..robots import RobotState, RoboyRobot
from .simulation_client import SimulationClient


class HipBatchSimulation:
    """N lock-step environments of one robot on one GPU."""

    def __init__(self, robot: RoboyRobot, n_envs: int, integrator="euler", step_size: float = 0.1,
                 n_substeps: int = 1, device: int = 0, seed: int = 0, env_id_offset: int = 0):
        self.robot = robot
        self._h = None
        self._lib = nat.load()
        self._desc = robot.get_description()
        if integrator not in nat.INTEGRATORS:
            raise ValueError("integrator must be 'euler' or 'rk4'")
        handle = ctypes.c_void_p()
        nat.check(self._lib.rb_create(
            ctypes.byref(self._desc.as_c_struct()), int(n_envs), nat.INTEGRATORS[integrator],
            float(step_size), int(n_substeps), int(device), int(seed), int(env_id_offset),
            ctypes.byref(handle)))
        self._h = handle
        self.n_envs = int(n_envs)
        self.n_q, self.n_t = self._desc.n_q, self._desc.n_t
        self.step_size = float(step_size)
        self._owned = []

    # -- lifetime ---------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None):
            for ptr in self._owned:
                self._lib.rb_free(self._h, ptr)
            self._owned = []
            self._lib.rb_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def handle(self):
        return self._h

    def info(self) -> dict:
        info = nat.SimInfo()
        nat.check(self._lib.rb_info(self._h, ctypes.byref(info)))
        return {name: getattr(info, name) for name, _ in nat.SimInfo._fields_}

    def dispatch(self, entry="step") -> dict:
        """The row of the library's dispatch table the next launch of `entry` ('step', 'env_step', 'fused_rollout') takes on this
        handle: ``{'id': 'ball8/step/env_per_lane/rk4/b256/table/v0', 'kernel': 1, 'block': 256, ...}``."""
        row = nat.DispatchRow()
        nat.check(self._lib.rb_dispatch_current(self._h, nat.ENTRIES[entry] if isinstance(entry, str) else int(entry), ctypes.byref(row)))
        return dict(row.as_dict(), id=row.row_id())

    def specialization(self) -> str:
        """'kernarg', 'table' (MsjRobot's ahead-of-time instances) or 'jit' (hiprtc instances on this robot's constants)."""
        return {0: "kernarg", 1: "table", 2: "jit"}[self._lib.rb_specialization(self._h)]

    def select_kernel(self, kernel: int):
        nat.check(self._lib.rb_select_kernel(self._h, int(kernel)))

    def set_stream(self, stream_ptr):
        """``None``: the handle's own stream; ``0``: the device's default (null) stream, which is
        what ``torch.cuda.current_stream().cuda_stream`` is unless another stream was made
        current; otherwise a ``hipStream_t`` value."""
        if stream_ptr is None:
            arg = ctypes.c_void_p(0)
        elif int(stream_ptr) == 0:
            arg = ctypes.c_void_p(nat.STREAM_DEVICE_DEFAULT)
        else:
            arg = ctypes.c_void_p(int(stream_ptr))
        nat.check(self._lib.rb_set_stream(self._h, arg))

    def synchronize(self):
        nat.check(self._lib.rb_synchronize(self._h))

    # -- host-array interface (numpy in, numpy out) -----------------------
    def _out(self):
        return (np.empty((self.n_envs, self.n_q), np.float32),
                np.empty((self.n_envs, self.n_q), np.float32),
                np.empty(self.n_envs, np.uint8))

    def forward_reset_command(self, mask=None):
        m = None if mask is None else np.ascontiguousarray(mask, dtype=np.uint8)
        nat.check(self._lib.rb_reset(self._h, nat.u8ptr(m)))
        return self.read_state()

    def read_state(self):
        q, qd, f = self._out()
        nat.check(self._lib.rb_read_state(self._h, nat.fptr(q), nat.fptr(qd), nat.u8ptr(f)))
        return q, qd, f.astype(bool)

    def set_state(self, q, qd, feasible=None):
        q = nat.as_f32(q, (self.n_envs, self.n_q), "q")
        qd = nat.as_f32(qd, (self.n_envs, self.n_q), "qd")
        f = None if feasible is None else np.ascontiguousarray(feasible, dtype=np.uint8)
        nat.check(self._lib.rb_set_state(self._h, nat.fptr(q), nat.fptr(qd), nat.u8ptr(f)))

    def forward_step_command(self, actions, act_scale: float = 1.0):
        """actions: [N, n_t] tendon set-points (``act_scale=1``) or raw policy
        actions in [-1, 1] (``act_scale`` = the robot's set-point bound)."""
        a = nat.as_f32(actions, (self.n_envs, self.n_t), "actions")
        q, qd, f = self._out()
        nat.check(self._lib.rb_step(self._h, nat.fptr(a), float(act_scale),
                                    nat.fptr(q), nat.fptr(qd), nat.u8ptr(f)))
        return q, qd, f.astype(bool)

    def get_new_goal_joint_angles(self, mask=None):
        m = None if mask is None else np.ascontiguousarray(mask, dtype=np.uint8)
        goals = np.empty((self.n_envs, self.n_q), np.float32)
        nat.check(self._lib.rb_sample_goals(self._h, nat.u8ptr(m), nat.fptr(goals)))
        return goals

    # -- device-pointer interface (no host copies) ------------------------
    def state_ptrs(self):
        q, qd, f = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p()
        nat.check(self._lib.rb_state_ptrs(self._h, ctypes.byref(q), ctypes.byref(qd), ctypes.byref(f)))
        return q.value, qd.value, f.value

    def malloc(self, nbytes: int) -> int:
        ptr = ctypes.c_void_p()
        nat.check(self._lib.rb_malloc(self._h, int(nbytes), ctypes.byref(ptr)))
        self._owned.append(ptr)
        return ptr.value

    def upload(self, d_ptr: int, array: np.ndarray):
        a = np.ascontiguousarray(array)
        nat.check(self._lib.rb_memcpy_h2d(self._h, ctypes.c_void_p(d_ptr),
                                          a.ctypes.data_as(ctypes.c_void_p), a.nbytes))

    def download(self, d_ptr: int, shape, dtype=np.float32) -> np.ndarray:
        out = np.empty(shape, dtype=dtype)
        nat.check(self._lib.rb_memcpy_d2h(self._h, out.ctypes.data_as(ctypes.c_void_p),
                                          ctypes.c_void_p(d_ptr), out.nbytes))
        return out

    def step_dev(self, d_act: int, act_scale: float = 1.0):
        nat.check(self._lib.rb_step_dev(self._h, ctypes.c_void_p(d_act), float(act_scale)))

    def range_capable(self, env_layer: bool = False) -> bool:
        """True if this handle's kernel form steps sub-ranges of the batch (``rb_range_capable``): the plain step, or
        (env_layer) the fused env step."""
        return bool(int(self._lib.rb_range_capable(self._h)) & (2 if env_layer else 1))

    def step_range_dev(self, first_env: int, n_envs: int, stream_ptr, d_act: int, act_scale: float = 1.0):
        """Envs [first_env, first_env + n_envs) on the stream ``stream_ptr`` (None / 0: the handle's); ``d_act`` is the WHOLE batch's slab."""
        nat.check(self._lib.rb_step_range_dev(self._h, int(first_env), int(n_envs), ctypes.c_void_p(int(stream_ptr or 0)),
                                              ctypes.c_void_p(d_act), float(act_scale)))

    def rollout_dev(self, d_act_ring: int, ring: int, n_steps: int, act_scale: float = 1.0,
                    use_graph: bool = False):
        nat.check(self._lib.rb_rollout_dev(self._h, ctypes.c_void_p(d_act_ring), int(ring),
                                           int(n_steps), float(act_scale), int(bool(use_graph))))

    def rollout_chains(self) -> int:
        """1: rollout_dev's graphs launch once per step over the whole batch; 2: the two halves step as two independent chains."""
        return int(self._lib.rb_rollout_chains(self._h))

    def set_rollout_chains(self, chains: int):
        """0: the library's choice; 1..4: that many chains of launches per step in graph rollouts (``rb_set_rollout_chains``)."""
        nat.check(self._lib.rb_set_rollout_chains(self._h, int(chains)))

    def rollout_fused_dev(self, d_act_ring: int, ring: int, n_steps: int, act_scale: float = 1.0):
        """Open-loop rollout in one launch (state in registers across the steps)."""
        nat.check(self._lib.rb_rollout_fused_dev(self._h, ctypes.c_void_p(d_act_ring), int(ring),
                                                 int(n_steps), float(act_scale)))

    def fill_actions_dev(self, d_act: int, step: int):
        nat.check(self._lib.rb_fill_actions_dev(self._h, ctypes.c_void_p(d_act), int(step)))

    def sample_goals_dev(self, d_goal: int, d_mask: int = 0):
        nat.check(self._lib.rb_sample_goals_dev(self._h, ctypes.c_void_p(d_mask or 0),
                                                ctypes.c_void_p(d_goal)))


class HipSimulationClient(SimulationClient):
    """Single-env ``SimulationClient`` backed by a batch of one on the GPU."""

    def __init__(self, robot: RoboyRobot, process_idx: int = 1, timeout_secs: int = 2,
                 integrator="euler", n_substeps: int = 1, device: int = 0, seed: int = 0):
        # process_idx plays the role it has in the reference (which simulator
        # instance, ros_simulation_client.py:27-30): here, the global env id
        # that keys this env's random streams.  timeout_secs is accepted for
        # signature compatibility; an in-process call cannot time out.
        self.robot = robot
        self._timeout_secs = timeout_secs
        self._step_size = 0.1   # ros_simulation_client.py:22
        self._sim = HipBatchSimulation(robot, 1, integrator=integrator, step_size=self._step_size,
                                       n_substeps=n_substeps, device=device, seed=seed,
                                       env_id_offset=int(process_idx))
        self._n_t = robot.get_action_space().shape[0]

    def _state(self, q, qd, feasible) -> RobotState:
        # float64 arrays like the reference's (ROS float lists -> np.array)
        return self.robot.new_state(joint_angle=q[0].astype(np.float64),
                                    joint_vel=qd[0].astype(np.float64),
                                    is_feasible=bool(feasible[0]))

    def read_state(self) -> RobotState:
        return self._state(*self._sim.read_state())

    def forward_step_command(self, action) -> RobotState:
        action = np.asarray(action, dtype=np.float32)
        if action.shape != (self._n_t,):
            raise TypeError("action must be a sequence of %d floats" % self._n_t)
        return self._state(*self._sim.forward_step_command(action[None, :]))

    def forward_reset_command(self) -> RobotState:
        return self._state(*self._sim.forward_reset_command())

    def get_new_goal_joint_angles(self) -> np.ndarray:
        return self._sim.get_new_goal_joint_angles()[0].astype(np.float64)

    def close(self):
        self._sim.close()
