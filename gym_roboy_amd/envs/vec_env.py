"""``RoboyVecEnv``: N ``RoboyEnv``s advanced by one fused kernel per step.

The reference vectorises by running one OS process, one ROS client and one
simulator instance per env under stable_baselines' ``SubprocVecEnv``
(``/root/reference/gym_roboy/train_parallel.py:19-29``); every env step costs
two process boundaries plus ~150 us of Python.  Here the whole env layer of
``RoboyEnv.step`` (``roboy_env.py:51-70``: action rescale, simulator step,
observation, reward, done, goal resampling) runs on the GPU next to the
physics, for all envs at once, and ``auto_reset`` reproduces what the
``SubprocVecEnv`` worker does on ``done`` (``env.reset()``, i.e. simulator
reset, step counter back to 1, a fresh goal, and the reset observation is the
one returned).

``step`` takes actions ``[N, n_t]`` in ``[-1, 1]`` either as a numpy array
(copied to the device) or as a CUDA ``torch.Tensor`` (used in place), and
returns ``(obs [N, 3 n_q], reward [N], done [N], infos)`` of the same kind.
"""
import ctypes

import numpy as np

from .. import _native as nat
from .._gymcompat import spaces
from . import reward as rw
from .robots import RoboyRobot
from .simulations.hip_simulation_client import HipBatchSimulation


class RoboyVecEnv:

    def __init__(self, robot: RoboyRobot, num_envs: int, seed: int = 0,
                 joint_vel_penalty: bool = False,
                 is_agent_getting_bonus_for_reaching_goal: bool = True,
                 auto_reset: bool = True, integrator="euler", n_substeps: int = 1,
                 device: int = 0, env_id_offset: int = 0, max_episode_length: int = 400):
        self.robot = robot
        self.num_envs = int(num_envs)
        self.sim = HipBatchSimulation(robot, num_envs, integrator=integrator, n_substeps=n_substeps,
                                      device=device, seed=seed, env_id_offset=env_id_offset)
        angles, vels, acts = (robot.get_joint_angles_space(), robot.get_joint_vels_space(),
                              robot.get_action_space())
        self.n_q, self.n_t = self.sim.n_q, self.sim.n_t
        self.action_space = spaces.Box(low=-1, high=1, shape=acts.shape, dtype="float32")
        self.observation_space = spaces.Box(
            low=np.concatenate((angles.low, vels.low, angles.low)),
            high=np.concatenate((angles.high, vels.high, angles.high)), dtype="float32")
        # thresholds exactly as the reference forms them (roboy_env.py:24-25,127,130)
        max_dist_angle = rw.l2_distance(angles.low, angles.high)
        max_dist_vel = rw.l2_distance(vels.low, vels.high)
        cfg = nat.EnvConfig()
        cfg.joint_vel_penalty = int(bool(joint_vel_penalty))
        cfg.goal_bonus = int(bool(is_agent_getting_bonus_for_reaching_goal))
        cfg.max_episode_length = int(max_episode_length)
        cfg.auto_reset = int(bool(auto_reset))
        cfg.penalty_boundary = 1.0
        cfg.bonus_goal = 1000.0
        cfg.angle_lo, cfg.angle_hi = float(angles.low[0]), float(angles.high[0])
        cfg.vel_lo, cfg.vel_hi = float(vels.low[0]), float(vels.high[0])
        cfg.action_lo, cfg.action_hi = float(acts.low[0]), float(acts.high[0])
        cfg.goal_angle_tol = float(max_dist_angle / 200)
        cfg.goal_vel_tol = float(max_dist_vel / 5)
        for box in (angles, vels, acts):
            if not (np.all(box.low == box.low[0]) and np.all(box.high == box.high[0])):
                raise NotImplementedError("fused env layer expects uniform per-joint boxes")
        self._cfg = cfg
        nat.check(self.sim._lib.rb_env_configure(self.sim.handle, ctypes.byref(cfg)))
        n = self.num_envs
        self._d_act = self.sim.malloc(4 * n * self.n_t)
        self._d_obs = self.sim.malloc(4 * n * 3 * self.n_q)
        self._d_rew = self.sim.malloc(4 * n)
        self._d_done = self.sim.malloc(4 * n)
        self._pending_actions = None
        self._seed = int(seed)
        self._replayed_env_steps = 0.0
        self._stream = None            # the simulation's own stream

    # ------------------------------------------------------------------
    def reset(self):
        nat.check(self.sim._lib.rb_env_reset_dev(self.sim.handle, ctypes.c_void_p(self._d_obs)))
        self.sim.synchronize()
        return self.sim.download(self._d_obs, (self.num_envs, 3 * self.n_q))

    def set_goal(self, goal_q, step_num=None):
        """Overwrite every env's goal ``[N, n_q]`` (and episode step counter ``[N]``): the
        batched form of assigning ``RoboyEnv._goal_state`` / ``.step_num`` as the reference's
        tests do (``gym_roboy/envs/tests/test_roboy_env.py:62-66,172-176``)."""
        g = nat.as_f32(goal_q, (self.num_envs, self.n_q), "goal_q")
        sn = None
        if step_num is not None:
            sn = np.ascontiguousarray(step_num, dtype=np.uint32)
            if sn.shape != (self.num_envs,):
                raise ValueError("step_num must have shape (%d,)" % self.num_envs)
        nat.check(self.sim._lib.rb_env_set_goal(
            self.sim.handle, nat.fptr(g), None if sn is None else sn.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32))))

    def step(self, actions):
        if _is_cuda_tensor(actions):
            return self._step_torch(actions)
        a = nat.as_f32(actions, (self.num_envs, self.n_t), "actions")
        self.sim.upload(self._d_act, a)
        self.step_dev(self._d_act, self._d_obs, self._d_rew, self._d_done)
        self.sim.synchronize()
        n = self.num_envs
        return (self.sim.download(self._d_obs, (n, 3 * self.n_q)),
                self.sim.download(self._d_rew, (n,)),
                self.sim.download(self._d_done, (n,), np.uint32).astype(bool), [{}] * n)

    def step_dev(self, d_act, d_obs, d_rew, d_done):
        """Raw device-pointer form: asynchronous on the simulation's stream."""
        nat.check(self.sim._lib.rb_env_step_dev(
            self.sim.handle, ctypes.c_void_p(d_act), ctypes.c_void_p(d_obs),
            ctypes.c_void_p(d_rew), ctypes.c_void_p(d_done)))

    def step_range_dev(self, first_env, n_envs, stream_ptr, d_act, d_obs, d_rew, d_done):
        """``step_dev`` for envs [first_env, first_env + n_envs) on the stream ``stream_ptr`` (None: the simulation's); the
        pointers are those of the WHOLE batch's arrays.  Disjoint ranges may be stepped concurrently on different streams
        (``rb_env_step_range_dev``): how a closed-loop caller overlaps one half's launch gaps and memory phases with the
        other half's arithmetic (``gym_roboy_amd/ppo.py``)."""
        nat.check(self.sim._lib.rb_env_step_range_dev(
            self.sim.handle, int(first_env), int(n_envs), ctypes.c_void_p(int(stream_ptr or 0)), ctypes.c_void_p(d_act),
            ctypes.c_void_p(d_obs), ctypes.c_void_p(d_rew), ctypes.c_void_p(d_done)))

    def range_capable(self) -> bool:
        return self.sim.range_capable(env_layer=True)

    def _step_torch(self, actions):
        import torch
        n = self.num_envs
        if actions.dtype != torch.float32 or tuple(actions.shape) != (n, self.n_t) or not actions.is_contiguous():
            raise ValueError("actions must be a contiguous float32 CUDA tensor of shape (%d, %d)" % (n, self.n_t))
        # run on torch's current stream so the policy's kernels and the env
        # step are ordered without a host sync
        self.set_stream(torch.cuda.current_stream(actions.device).cuda_stream)
        obs = torch.empty((n, 3 * self.n_q), dtype=torch.float32, device=actions.device)
        rew = torch.empty((n,), dtype=torch.float32, device=actions.device)
        done = torch.empty((n,), dtype=torch.int32, device=actions.device)
        self.step_dev(actions.data_ptr(), obs.data_ptr(), rew.data_ptr(), done.data_ptr())
        return obs, rew, done.bool(), [{}] * n

    # -- the rest of stable_baselines' VecEnv surface (what PPO2 / wrappers call on the
    #    reference's SubprocVecEnv, train_parallel.py:29) ---------------------------------
    def step_async(self, actions):
        self._pending_actions = actions

    def step_wait(self):
        actions, self._pending_actions = self._pending_actions, None
        if actions is None:
            raise RuntimeError("step_wait() without step_async()")
        return self.step(actions)

    def seed(self, seed=None):
        """The random streams are keyed at construction (seed, global env id); a VecEnv
        cannot be re-seeded in place.  Returns one entry per env like SubprocVecEnv."""
        if seed is not None and int(seed) != self._seed:
            raise NotImplementedError("construct RoboyVecEnv(seed=%d) instead of re-seeding" % int(seed))
        return [None] * self.num_envs

    def get_attr(self, attr_name, indices=None):
        n = self.num_envs if indices is None else len(list(indices))
        return [getattr(self, attr_name)] * n

    def env_method(self, method_name, *args, indices=None, **kwargs):
        raise NotImplementedError("the envs of a RoboyVecEnv are not separate Python objects (no %s)" % method_name)

    def render(self, mode="human"):
        pass        # RoboyEnv.render is a no-op too (roboy_env.py:89-90)

    def set_stream(self, stream_ptr):
        """Stream of every later launch of this env (``HipBatchSimulation.set_stream``: 0 = the
        device's default stream).  A change drains the previous stream, so it is only forwarded
        when the stream really changes."""
        if stream_ptr != self._stream:
            self.sim.set_stream(stream_ptr)
            self._stream = stream_ptr

    def stats(self, reset: bool = False) -> dict:
        out = (ctypes.c_double * 8)()
        nat.check(self.sim._lib.rb_env_stats(self.sim.handle, out, int(reset)))
        keys = ("sum_return", "sum_return_sq", "n_episodes", "sum_length", "n_goal_reached",
                "n_infeasible_steps", "n_env_steps", "sum_reward")
        stats = dict(zip(keys, list(out)))
        stats["n_env_steps"] += self._replayed_env_steps    # steps replayed from a captured graph
        if reset:
            self._replayed_env_steps = 0.0
        return stats

    def note_replayed_steps(self, n_steps: int):
        """n_env_steps is counted where launches are issued; a caller that replays a
        captured graph of `n_steps` env steps (ppo.py) reports them here."""
        self._replayed_env_steps += float(n_steps) * self.num_envs

    def stats_dev(self, d_out8: int, reset: bool = False):
        nat.check(self.sim._lib.rb_env_stats_dev(self.sim.handle, ctypes.c_void_p(d_out8), int(reset)))

    def close(self):
        self.sim.close()


def _is_cuda_tensor(x):
    return type(x).__module__.startswith("torch") and getattr(x, "is_cuda", False)
