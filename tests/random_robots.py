"""Seeded random joint-tree robots (roboy-tendon-robot/1 descriptions) for the parity tests of the generic
kernel: random topology (chains, bushes, several roots), axes, origins, inertial data (some links massless),
tendons routed over random links (base included; same-link segments; tendons that never leave one link)."""
import math

import numpy as np


def random_tree_spec(seed, n_q=None, n_t=None, shape=None, p_massless=0.25):
    """shape: None (random mix), "chain" (one serial chain), "star" (every joint a child of joint 0)."""
    from gym_roboy_amd.envs.robots.description import FORMAT_TAG
    rng = np.random.default_rng(1000 + seed)
    n_q = int(rng.integers(1, 25)) if n_q is None else n_q
    n_t = int(rng.integers(1, min(2 * n_q + 2, 40))) if n_t is None else n_t
    p_chain = rng.uniform(0.2, 0.9)
    joints = []
    for i in range(n_q):
        if i == 0:
            parent = -1
        elif shape == "chain":
            parent = i - 1
        elif shape == "star":
            parent = 0
        elif rng.random() < p_chain:
            parent = i - 1
        else:
            parent = int(rng.integers(-1, i))            # -1: another root
        axis = rng.normal(size=3)
        axis /= np.linalg.norm(axis)
        massless = rng.random() < p_massless
        mass = 0.0 if massless else float(rng.uniform(0.05, 0.8))
        if massless:
            inertia = [0.0] * 6
        else:
            a = rng.normal(size=(3, 3)) * 0.02
            m = a @ a.T * mass + np.eye(3) * 2e-4
            inertia = [m[0, 0], m[1, 1], m[2, 2], m[0, 1], m[0, 2], m[1, 2]]
        lim = float(rng.uniform(0.3, 1.0))
        joints.append({"name": "j%d" % i, "parent": parent, "axis": [float(x) for x in axis],
                       "origin": [float(x) for x in rng.uniform(-0.08, 0.08, 3)], "mass": mass,
                       "com": [float(x) for x in rng.uniform(-0.03, 0.03, 3)], "inertia": [float(x) for x in inertia],
                       "armature": float(rng.uniform(0.03, 0.1)), "damping": float(rng.uniform(0.1, 0.5)),
                       "limit": [-lim * float(rng.uniform(0.5, 1.0)), lim],
                       "max_velocity": float(rng.uniform(math.pi / 8, math.pi / 3))})
    tendons = []
    for k in range(n_t):
        n_vp = int(rng.integers(2, 6))
        links, pts = [], []
        link = int(rng.integers(-1, n_q))
        for v in range(n_vp):
            if v and rng.random() < 0.7:                  # move on to another link (or stay: a same-link segment)
                link = int(rng.integers(-1, n_q))
            links.append(link)
            pts.append(rng.uniform(-0.06, 0.06, 3) + (0.0 if link >= 0 else rng.uniform(-0.1, 0.1, 3)))
        tendons.append({"name": "t%d" % k, "f_max": float(rng.uniform(4.0, 25.0)),
                        "via_points": [{"link": l, "pos": [float(x) for x in p]} for l, p in zip(links, pts)]})
    return {"format": FORMAT_TAG, "name": "random%d" % seed, "gravity": [0.0, 0.0, -9.81], "joints": joints,
            "tendons": tendons,
            "muscle": {"kp": 10.0, "setpoint_scale": 0.02, "v_max": 8.0, "fl_width": 0.45, "kpe": 4.0, "e0": 0.6,
                       "fv_a": 0.25, "fv_n": 1.5}}


def random_tree_robot(seed, **kw):
    """(robot object, description) of random_tree_spec(seed)."""
    from gym_roboy_amd._gymcompat import spaces
    from gym_roboy_amd.envs.robots import RobotDescription
    from gym_roboy_amd.envs.robots.roboy_robot import RoboyRobot
    desc = RobotDescription(random_tree_spec(seed, **kw))

    class Random(RoboyRobot):
        _DIM_JOINT_ANGLE, _DIM_ACTION = desc.n_q, desc.n_t
        _JOINT_ANGLE_SPACE = spaces.Box(low=-math.pi, high=math.pi, shape=(desc.n_q,), dtype="float32")
        _JOINT_VEL_SPACE = spaces.Box(low=-math.pi / 3, high=math.pi / 3, shape=(desc.n_q,), dtype="float32")
        _ACTION_SPACE = spaces.Box(low=-0.3, high=0.3, shape=(desc.n_t,), dtype="float32")

        @classmethod
        def get_description(cls):
            return desc
    return Random(), desc
