#!/usr/bin/env python3
"""Generate tests/golden/env_layer.json from the REFERENCE's own Python.

Runs only in the build container (needs /root/reference); the fixture it
writes is plain data (inputs and the reference's outputs) and is what travels.
The reference imports gym, typeguard, rclpy and roboy_simulation_msgs, none of
which is installed here, so this script first writes four tiny stand-in
modules into a temporary directory (the minimum the imported lines touch:
``spaces.Box``, ``Env``/``GoalEnv``, ``register``, an identity ``typechecked``
and empty ROS names) and puts it on ``sys.path``.  Inputs are explicit arrays
from a seeded ``numpy.random.default_rng`` stored in the fixture, never
``Box.sample()`` (the stand-in's sampling is not gym's).

    python tests/golden/make_env_golden.py
"""
import json
import os
import sys
import tempfile
import textwrap

sys.dont_write_bytecode = True
REFERENCE = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "env_layer.json")

STANDINS = {
    "gym/__init__.py": """
        from . import spaces
        class Env:
            reward_range = (-float("inf"), float("inf"))   # as gym.Env defines it
        class GoalEnv(Env):
            pass
    """,
    "gym/spaces.py": """
        import numpy as np
        class Box:
            def __init__(self, low, high, shape=None, dtype="float32"):
                self.dtype = np.dtype(dtype)
                if shape is None:
                    self.low = np.asarray(low).astype(self.dtype)
                    self.high = np.asarray(high).astype(self.dtype)
                    self.shape = self.low.shape
                else:
                    self.shape = tuple(shape)
                    self.low = np.full(self.shape, low, dtype=self.dtype)
                    self.high = np.full(self.shape, high, dtype=self.dtype)
            def sample(self):
                return np.random.uniform(self.low, self.high, self.shape).astype(self.dtype)
            def contains(self, x):
                x = np.asarray(x)
                return bool(x.shape == self.shape and np.all(x >= self.low) and np.all(x <= self.high))
    """,
    "gym/envs/__init__.py": "",
    "gym/envs/registration.py": """
        registry = {}
        def register(id, entry_point, **kw):
            registry[id] = entry_point
    """,
    "typeguard.py": """
        def typechecked(f):
            return f
    """,
    "rclpy.py": """
        def init():
            pass
        class _Node:
            def create_client(self, *a, **k):
                return object()
        def create_node(name):
            return _Node()
    """,
    "roboy_simulation_msgs/__init__.py": "",
    "roboy_simulation_msgs/srv.py": """
        class _Srv:
            class Request:
                pass
        GymStep = GymReset = GymGoal = _Srv
    """,
}


def main():
    tmp = tempfile.mkdtemp(prefix="refshim_")
    for rel, body in STANDINS.items():
        path = os.path.join(tmp, rel)
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, "w") as fh:
            fh.write(textwrap.dedent(body))
    sys.path[:0] = [tmp, REFERENCE]

    import numpy as np
    import gym_roboy  # noqa: F401  (the reference; registers msj-control-v0)
    from gym.envs.registration import registry
    from gym_roboy.envs import RoboyEnv
    from gym_roboy.envs.roboy_env import _l2_distance, _rescale_from_one_space_to_other
    from gym_roboy.envs.robots import MsjRobot, RobotState
    from gym_roboy.envs.simulations import SimulationClient, StubSimulationClient
    from gym import spaces

    rng = np.random.default_rng(20261003)
    robot = MsjRobot()
    A, V, U = robot.get_joint_angles_space(), robot.get_joint_vels_space(), robot.get_action_space()
    L = lambda a: np.asarray(a, dtype=np.float64).tolist()  # noqa: E731

    class ScriptedClient(SimulationClient):
        """Plays back a fixed list of states (float64 lists, like ROS would deliver)."""
        def __init__(self, states, goals):
            self.robot = robot
            self.states, self.goals = list(states), list(goals)
            self.i = self.g = 0
            self.received = []
        def read_state(self):
            return robot.new_state(joint_angle=[0.0] * 3, joint_vel=[0.0] * 3, is_feasible=True)
        def forward_step_command(self, action):
            self.received.append(list(action))
            q, qd, ok = self.states[self.i]
            self.i += 1
            return robot.new_state(joint_angle=list(q), joint_vel=list(qd), is_feasible=bool(ok))
        def forward_reset_command(self):
            return self.read_state()
        def get_new_goal_joint_angles(self):
            g = self.goals[self.g % len(self.goals)]
            self.g += 1
            return np.array(g)

    fx = {"generated_by": "tests/golden/make_env_golden.py from /root/reference (Roboy/gym-roboy)",
          "registry_ids": sorted(registry)}

    # --- robot constants (msj_robot.py:8-16) ---
    fx["spaces"] = {"angle_low": L(A.low), "angle_high": L(A.high), "vel_low": L(V.low), "vel_high": L(V.high),
                    "action_low": L(U.low), "action_high": L(U.high),
                    "dtype": [str(A.low.dtype), str(V.low.dtype), str(U.low.dtype)]}

    # --- env constants and reward ranges (roboy_env.py:24-38) ---
    envs = {}
    for pen in (False, True):
        for bonus in (False, True):
            env = RoboyEnv(simulation_client=StubSimulationClient(robot=robot), joint_vel_penalty=pen,
                           is_agent_getting_bonus_for_reaching_goal=bonus)
            envs[(pen, bonus)] = env
    e0 = envs[(False, True)]
    fx["env_constants"] = {
        "max_distance_joint_angle": float(e0._MAX_DISTANCE_JOINT_ANGLE),
        "max_distance_joint_vels": float(e0._MAX_DISTANCE_JOINT_VELS),
        "penalty": e0._PENALTY_FOR_TOUCHING_BOUNDARY, "bonus": e0._BONUS_FOR_REACHING_GOAL,
        "max_episode_length": e0._MAX_EPISODE_LENGTH, "initial_step_num": e0.step_num,
        "goal_joint_vel": L(e0._GOAL_JOINT_VEL),
        "obs_low": L(e0.observation_space.low), "obs_high": L(e0.observation_space.high),
        "obs_dtype": str(e0.observation_space.low.dtype),
        "action_space_low": L(e0.action_space.low), "action_space_high": L(e0.action_space.high),
    }
    fx["reward_range"] = [{"joint_vel_penalty": p, "bonus": b, "range": [float(x) for x in envs[(p, b)].reward_range]}
                          for (p, b) in envs]

    # --- rescale (roboy_env.py:143-158) ---
    cases = [np.linspace(-1, 1, 8).astype(np.float32), np.zeros(8, np.float32),
             -np.ones(8, np.float32), np.ones(8, np.float32)]
    cases += [rng.uniform(-1, 1, 8).astype(np.float32) for _ in range(6)]
    fx["rescale"] = [{"input": L(x), "output": L(_rescale_from_one_space_to_other(
        input_val=x, input_space=e0.action_space, output_space=U)),
        "output_dtype": str(_rescale_from_one_space_to_other(input_val=x, input_space=e0.action_space,
                                                             output_space=U).dtype)} for x in cases]
    lo, hi = -rng.uniform(size=8), rng.uniform(size=8)
    lo2, hi2 = -rng.uniform(size=8), rng.uniform(size=8)
    sp_in, sp_out = spaces.Box(low=lo, high=hi, dtype="float32"), spaces.Box(low=lo2, high=hi2, dtype="float32")
    x = (sp_in.low + (sp_in.high - sp_in.low) * rng.uniform(size=8).astype(np.float32)).astype(np.float32)
    x = np.clip(x, sp_in.low, sp_in.high)
    fx["rescale_general"] = {"in_low": L(sp_in.low), "in_high": L(sp_in.high), "out_low": L(sp_out.low),
                             "out_high": L(sp_out.high), "input": L(x),
                             "output": L(_rescale_from_one_space_to_other(input_val=x, input_space=sp_in,
                                                                          output_space=sp_out))}

    # --- normalisation, interpolation (roboy_robot.py:12-18,80-95) ---
    norm = []
    for _ in range(6):
        q, qd = rng.uniform(-3.1, 3.1, 3), rng.uniform(-0.5, 0.5, 3)
        st = robot.normalize_state(RobotState(joint_angles=q, joint_vels=qd, is_feasible=True))
        norm.append({"q": L(q), "qd": L(qd), "q_norm": L(st.joint_angles), "qd_norm": L(st.joint_vels)})
    fx["normalize"] = norm
    mx, mn = robot.normalize_state(robot.new_max_state()), robot.normalize_state(robot.new_min_state())
    fx["normalize_corners"] = {"max_q": L(mx.joint_angles), "max_qd": L(mx.joint_vels), "max_feasible": mx.is_feasible,
                               "min_q": L(mn.joint_angles), "min_qd": L(mn.joint_vels)}
    s1 = RobotState(rng.uniform(-1, 1, 3), rng.uniform(-1, 1, 3), True)
    s2 = RobotState(rng.uniform(-1, 1, 3), rng.uniform(-1, 1, 3), False)
    mid = RobotState.interpolate(s1, s2)
    fx["interpolate"] = {"a": [L(s1.joint_angles), L(s1.joint_vels), s1.is_feasible],
                         "b": [L(s2.joint_angles), L(s2.joint_vels), s2.is_feasible],
                         "mid": [L(mid.joint_angles), L(mid.joint_vels), mid.is_feasible]}
    fx["l2_distance"] = {"inf_minus_inf": float(_l2_distance(np.array([np.inf, 1.0]), np.array([np.inf, 0.0])))}

    # --- compute_reward / _did_reach_goal (roboy_env.py:92-134) ---
    import contextlib
    import io
    rewards = []
    thr_a, thr_v = e0._MAX_DISTANCE_JOINT_ANGLE / 200, e0._MAX_DISTANCE_JOINT_VELS / 5
    states = [(np.array([0.1, -0.2, 0.3]), np.array([0.01, 0.02, -0.03]), np.array([0.5, 0.4, -0.3]))]
    for _ in range(24):
        states.append((rng.uniform(-3.0, 3.0, 3), rng.uniform(-0.5, 0.5, 3), rng.uniform(-3.0, 3.0, 3)))
    g = rng.uniform(-2, 2, 3)
    d = rng.normal(size=3); d /= np.linalg.norm(d)
    w = rng.normal(size=3); w /= np.linalg.norm(w)
    for fa in (0.0, 0.5, 0.999, 1.001, 3.0):       # either side of the angle threshold
        for fv in (0.0, 0.999, 1.001):               # either side of the velocity threshold
            states.append((g + fa * thr_a * d, fv * thr_v * w, g))
    for (q, qd, goal) in states:
        for feasible in (True, False):
            cur = RobotState(joint_angles=q, joint_vels=qd, is_feasible=feasible)
            gs = robot.new_state(joint_angle=goal, joint_vel=e0._GOAL_JOINT_VEL, is_feasible=True)
            with contextlib.redirect_stdout(io.StringIO()):
                reached = e0._did_reach_goal(current_state=cur, goal_state=gs)
                row = {"q": L(q), "qd": L(qd), "goal_q": L(goal), "feasible": feasible, "reached": bool(reached),
                       "reward": {}}
                for (p, b), env in envs.items():
                    row["reward"]["pen%d_bonus%d" % (p, b)] = env.compute_reward(current_state=cur, goal_state=gs)
            rewards.append(row)
    fx["reward_cases"] = rewards
    fx["goal_thresholds"] = {"angle": float(thr_a), "vel": float(thr_v)}

    # --- a scripted episode through RoboyEnv.step (roboy_env.py:51-70) ---
    T = 12
    script = []
    goals = [rng.uniform(-2, 2, 3).tolist() for _ in range(6)]
    for t in range(T):
        script.append((rng.uniform(-3, 3, 3).tolist(), rng.uniform(-0.5, 0.5, 3).tolist(), bool(t % 5 != 3)))
    script[7] = (goals[1], [0.0, 0.0, 0.0], True)   # lands on the current goal -> bonus, done, new goal
    actions = [rng.uniform(-1, 1, 8).astype(np.float32) for _ in range(T)]
    client = ScriptedClient(script, goals)
    env = RoboyEnv(simulation_client=client)      # draws goals[0]
    with contextlib.redirect_stdout(io.StringIO()):
        obs0 = env.reset()                            # draws goals[1]
        steps = []
        for t in range(T):
            obs, rew, done, info = env.step(actions[t])
            steps.append({"obs": L(obs), "obs_dtype": str(obs.dtype), "reward": rew, "done": done,
                          "step_num": env.step_num, "goal_after": L(env._goal_state.joint_angles)})
    fx["episode"] = {"script": script, "goals": goals, "actions": [L(a) for a in actions],
                     "reset_obs": L(obs0), "received_setpoints": client.received, "steps": steps,
                     "reward_type": type(steps[0]["reward"]).__name__, "done_type": type(steps[0]["done"]).__name__}

    # --- episode length (test_roboy_env.py:170-180) ---
    env = RoboyEnv(simulation_client=ScriptedClient([([0.1, 0.1, 0.1], [0.0] * 3, True)] * 4, goals))
    env.reset()
    env.step_num = env._MAX_EPISODE_LENGTH - 1
    _, _, d1, _ = env.step(np.zeros(8, np.float32))
    _, _, d2, _ = env.step(np.zeros(8, np.float32))
    _, _, d3, _ = env.step(np.zeros(8, np.float32))
    fx["episode_length"] = {"done_at_399_plus_1": d1, "done_at_400_plus_1": d2, "done_after": d3,
                            "step_num_after": env.step_num}

    with open(OUT, "w") as fh:
        json.dump(fx, fh, indent=1)
        fh.write("\n")
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
