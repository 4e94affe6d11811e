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


def random_mirrored_spec(seed, n_branch=4, n_t_branch=5, extra_cross=False):
    """A trunk of two joints carrying TWO structurally identical branches (same axes, same massless links, same tendon
    routing) whose constants differ - what the lane-kernel generator writes as one stream of pair values - plus a plain
    third branch and a few trunk tendons.  extra_cross: one tendon runs from the first branch to the second (no pairing then)."""
    base = random_tree_spec(seed, n_q=2 + n_branch, n_t=1, shape="chain")
    rng = np.random.default_rng(7000 + seed)
    joints = base["joints"][:2]
    branch = base["joints"][2:]
    first = len(joints)

    def jitter(x, rel=0.2):
        return float(x * (1.0 + rng.uniform(-rel, rel)))

    for copy in range(2):
        off = len(joints) - 2
        for j, src in enumerate(branch):
            jj = dict(src)
            jj["name"] = "b%d_%d" % (copy, j)
            jj["parent"] = 1 if j == 0 else src["parent"] + off
            if copy == 1:                                   # the same structure, other numbers (zeros stay zeros: massless links)
                jj["origin"] = [jitter(x) for x in src["origin"]]
                jj["com"] = [jitter(x) for x in src["com"]]
                jj["mass"] = jitter(src["mass"])
                jj["inertia"] = [jitter(x, 0.05) for x in src["inertia"]]
                jj["armature"], jj["damping"] = jitter(src["armature"]), jitter(src["damping"])
                jj["limit"] = [jitter(src["limit"][0]), jitter(src["limit"][1])]
            joints.append(jj)
    second = first + n_branch
    third = len(joints)
    extra = random_tree_spec(seed + 50, n_q=3, n_t=1, shape="chain")["joints"]
    for j, src in enumerate(extra):
        jj = dict(src)
        jj["name"] = "c%d" % j
        jj["parent"] = 1 if j == 0 else third + j - 1
        joints.append(jj)
    tendons = []
    for k in range(n_t_branch):                              # routed over the trunk (or the base) and the first branch; mirrored
        n_vp = int(rng.integers(2, 5))
        links, pts = [], []
        for v in range(n_vp):
            links.append(int(rng.choice([-1, 0, 1] + list(range(first, first + n_branch)))))
            pts.append(rng.uniform(-0.06, 0.06, 3))
        if all(l < first for l in links):
            links[-1] = first + int(rng.integers(0, n_branch))
        f = float(rng.uniform(4.0, 25.0))
        tendons.append({"name": "a%d" % k, "f_max": f, "via_points": [{"link": l, "pos": [float(x) for x in p]} for l, p in zip(links, pts)]})
        tendons.append({"name": "b%d" % k, "f_max": jitter(f), "via_points": [{"link": l + n_branch if l >= first else l, "pos": [jitter(x) for x in p]}
                                                                                for l, p in zip(links, pts)]})
    for k in range(3):                                       # plain tendons: trunk, base, third branch
        links = [int(rng.choice([-1, 0, 1, third, third + 1, third + 2])) for _ in range(3)]
        tendons.append({"name": "p%d" % k, "f_max": float(rng.uniform(4.0, 25.0)),
                        "via_points": [{"link": l, "pos": [float(x) for x in rng.uniform(-0.06, 0.06, 3)]} for l in links]})
    if extra_cross:
        tendons.append({"name": "x", "f_max": 10.0, "via_points": [{"link": first, "pos": [0.01, 0.02, 0.03]}, {"link": second, "pos": [0.02, -0.01, 0.03]}]})
    rng.shuffle(tendons)
    spec = dict(base)
    spec["name"] = "mirrored%d" % seed
    spec["joints"], spec["tendons"] = joints, tendons
    return spec


def random_ball_joint_spec(seed, n_t=8):
    """One rigid body on an x-y-z ball joint at the base origin (the class msj_math.hpp closes in closed form) with
    everything else random: tendon routing (1-3 base via-points, 1-2 body via-points), inertia with products,
    centre of mass off the axis, tilted gravity, asymmetric limits, muscle parameters."""
    from gym_roboy_amd.envs.robots.description import FORMAT_TAG
    rng = np.random.default_rng(5000 + seed)

    def joint(name, parent, axis, **kw):
        lim = float(rng.uniform(0.3, 0.6))
        j = {"name": name, "parent": parent, "axis": axis, "origin": [0.0, 0.0, 0.0], "mass": 0.0, "com": [0.0, 0.0, 0.0],
             "inertia": [0.0] * 6, "armature": float(rng.uniform(0.1, 0.3)), "damping": float(rng.uniform(0.3, 1.0)),
             "limit": [-lim * float(rng.uniform(0.6, 1.0)), lim], "max_velocity": float(rng.uniform(math.pi / 8, math.pi / 4))}
        j.update(kw)
        return j
    simple = seed % 3 == 0                                  # every third robot takes the principal-axis fast path
    mass = float(rng.uniform(0.1, 0.5))
    a = rng.normal(size=(3, 3)) * 0.02
    m = a @ a.T * mass + np.eye(3) * 3e-4
    inertia = [m[0, 0], m[1, 1], m[2, 2], 0.0, 0.0, 0.0] if simple else [m[0, 0], m[1, 1], m[2, 2], m[0, 1], m[0, 2], m[1, 2]]
    com = [0.0, 0.0, float(rng.uniform(0.03, 0.08))] if simple else [float(x) for x in rng.uniform(-0.03, 0.06, 3)]
    gravity = [0.0, 0.0, -9.81] if simple else [float(x) for x in rng.normal(size=3) * 3.0 + np.array([0.0, 0.0, -9.0])]
    tendons = []
    for k in range(n_t):
        ang = rng.uniform(0, 2 * math.pi)
        pts = []
        for v in range(int(rng.integers(1, 4))):            # base side, from the motor towards the joint
            r, z = 0.13 - 0.02 * v + rng.uniform(-0.01, 0.01), -0.10 + 0.04 * v + rng.uniform(-0.01, 0.01)
            pts.append({"link": -1, "pos": [float(r * math.cos(ang)), float(r * math.sin(ang)), float(z)]})
        ang2 = ang + rng.uniform(-0.6, 0.6)
        for v in range(int(rng.integers(1, 3))):            # body side
            r, z = 0.07 - 0.02 * v + rng.uniform(-0.01, 0.01), 0.08 + 0.03 * v + rng.uniform(-0.01, 0.01)
            pts.append({"link": 2, "pos": [float(r * math.cos(ang2)), float(r * math.sin(ang2)), float(z)]})
        tendons.append({"name": "t%d" % k, "f_max": float(rng.uniform(4.0, 30.0)), "via_points": pts})
    return {"format": FORMAT_TAG, "name": "ball%d" % seed, "gravity": gravity,
            "joints": [joint("x", -1, [1.0, 0.0, 0.0]), joint("y", 0, [0.0, 1.0, 0.0]),
                       joint("z", 1, [0.0, 0.0, 1.0], mass=mass, com=com, inertia=[float(x) for x in inertia])],
            "tendons": tendons,
            "muscle": {"kp": float(rng.uniform(5.0, 15.0)), "setpoint_scale": float(rng.uniform(0.05, 0.15)),
                       "v_max": float(rng.uniform(4.0, 10.0)), "fl_width": float(rng.uniform(0.3, 0.6)),
                       "kpe": float(rng.uniform(3.0, 5.0)), "e0": float(rng.uniform(0.4, 0.8)),
                       "fv_a": float(rng.uniform(0.2, 0.35)), "fv_n": float(rng.uniform(1.3, 1.8))}}


def _robot_of(spec):
    from gym_roboy_amd._gymcompat import spaces
    from gym_roboy_amd.envs.robots import RobotDescription
    from gym_roboy_amd.envs.robots.roboy_robot import RoboyRobot
    desc = RobotDescription(spec)

    class Random(RoboyRobot):
        _DIM_JOINT_ANGLE, _DIM_ACTION = desc.n_q, desc.n_t
        _JOINT_ANGLE_SPACE = spaces.Box(low=-math.pi, high=math.pi, shape=(desc.n_q,), dtype="float32")
        _JOINT_VEL_SPACE = spaces.Box(low=-math.pi / 3, high=math.pi / 3, shape=(desc.n_q,), dtype="float32")
        _ACTION_SPACE = spaces.Box(low=-0.3, high=0.3, shape=(desc.n_t,), dtype="float32")

        @classmethod
        def get_description(cls):
            return desc
    return Random(), desc


def random_ball_joint_robot(seed, n_t=8):
    """(robot object, description) of random_ball_joint_spec(seed, n_t)."""
    return _robot_of(random_ball_joint_spec(seed, n_t))


def random_tree_robot(seed, **kw):
    """(robot object, description) of random_tree_spec(seed)."""
    return _robot_of(random_tree_spec(seed, **kw))


def random_mirrored_robot(seed, **kw):
    """(robot object, description) of random_mirrored_spec(seed)."""
    return _robot_of(random_mirrored_spec(seed, **kw))
