"""``UpperBodyRobot``: a synthetic 20-DOF / 38-tendon Roboy-like upper body.

The reference has no such class: ``RoboyRobot`` is abstract, ``MsjRobot`` the
only concrete robot, and its README only mentions an "Upper Body" as something
the framework could control (``/root/reference/README.md:6-7``,
``gym_roboy/envs/robots/roboy_robot.py:21-33``, ``TODO.md:1``).  BASELINE.json
nevertheless lists "Roboy upper-body RoboyRobot (~20 DOF, ~38 tendons)" as a
configuration, so this module defines one in the ``roboy-tendon-robot/1``
format: spine (3 revolutes), neck (3), and two 7-DOF arms (shoulder 3, elbow
1, forearm roll 1, wrist 2) = 20 joints; 6 + 6 + 2 x 13 = 38 tendons routed
over 3-4 via-points each, generated from a fixed seed and committed as
``data/upper_body.json``.  It is a benchmark geometry, not a model of the real
Roboy 2.0 (whose CARDSflow model files are not available).

Zero pose: torso and neck upright, arms hanging, so every link's centre of
mass lies on the vertical through its joint and the zero pose with zero
set-points is an equilibrium, as for ``MsjRobot``.  The boxes follow
``MsjRobot``'s conventions (angles +-pi, velocities +-pi/6, set-points +-0.3).
"""
import math
import os

import numpy as np

from ..._gymcompat import spaces
from .description import FORMAT_TAG, RobotDescription
from .roboy_robot import RoboyRobot

_DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "upper_body.json")


def upper_body_spec(seed: int = 2026) -> dict:
    """Generate the description (deterministic for a given seed)."""
    rng = np.random.default_rng(seed)
    joints = []

    def add(name, parent, axis, origin, mass, com, inertia, limit, armature=0.25, damping=1.0):
        joints.append({"name": name, "parent": parent, "axis": axis, "origin": origin, "mass": mass,
                       "com": com, "inertia": inertia, "armature": armature, "damping": damping,
                       "limit": [-limit, limit], "max_velocity": math.pi / 6})
        return len(joints) - 1

    def box_inertia(m, sx, sy, sz):
        return [m * (sy * sy + sz * sz) / 12, m * (sx * sx + sz * sz) / 12, m * (sx * sx + sy * sy) / 12, 0.0, 0.0, 0.0]

    zero3, none6 = [0.0, 0.0, 0.0], [0.0] * 6
    # spine: three co-located revolutes, the torso hangs on the last one
    s0 = add("spine_x", -1, [1, 0, 0], [0, 0, 0.0], 0.0, zero3, none6, 0.35, armature=0.6, damping=2.0)
    s1 = add("spine_y", s0, [0, 1, 0], zero3, 0.0, zero3, none6, 0.35, armature=0.6, damping=2.0)
    torso = add("spine_z", s1, [0, 0, 1], zero3, 6.0, [0, 0, 0.20], box_inertia(6.0, 0.30, 0.18, 0.40), 0.5,
                armature=0.6, damping=2.0)
    # neck on top of the torso
    n0 = add("neck_x", torso, [1, 0, 0], [0, 0, 0.42], 0.0, zero3, none6, 0.4)
    n1 = add("neck_y", n0, [0, 1, 0], zero3, 0.0, zero3, none6, 0.4)
    head = add("neck_z", n1, [0, 0, 1], zero3, 1.2, [0, 0, 0.08], box_inertia(1.2, 0.16, 0.18, 0.20), 0.6)
    arms = {}
    for side, sy in (("left", 1.0), ("right", -1.0)):
        sh0 = add(side + "_shoulder_x", torso, [1, 0, 0], [0, sy * 0.20, 0.36], 0.0, zero3, none6, 0.7)
        sh1 = add(side + "_shoulder_y", sh0, [0, 1, 0], zero3, 0.0, zero3, none6, 0.7)
        upper = add(side + "_shoulder_z", sh1, [0, 0, 1], zero3, 1.6, [0, 0, -0.13],
                    box_inertia(1.6, 0.07, 0.07, 0.28), 0.7)
        elbow = add(side + "_elbow", upper, [0, 1, 0], [0, 0, -0.28], 0.9, [0, 0, -0.06],
                    box_inertia(0.9, 0.06, 0.06, 0.12), 0.9)
        roll = add(side + "_forearm_roll", elbow, [0, 0, 1], [0, 0, -0.12], 0.5, [0, 0, -0.06],
                   box_inertia(0.5, 0.05, 0.05, 0.12), 0.8)
        w0 = add(side + "_wrist_x", roll, [1, 0, 0], [0, 0, -0.12], 0.0, zero3, none6, 0.5, armature=0.15, damping=0.6)
        hand = add(side + "_wrist_y", w0, [0, 1, 0], zero3, 0.4, [0, 0, -0.05], box_inertia(0.4, 0.08, 0.03, 0.10), 0.5,
                   armature=0.15, damping=0.6)
        arms[side] = (upper, elbow, roll, hand, sy)

    tendons = []

    def ring_point(radius, angle, z, jitter=0.004):
        return [float(radius * math.cos(angle) + rng.normal(0, jitter)),
                float(radius * math.sin(angle) + rng.normal(0, jitter)), float(z + rng.normal(0, jitter))]

    def add_ring(name, count, base_link, base_r, base_z, tip_link, tip_r, tip_z, f_max, shift=(0.0, 0.0), twist=0.5):
        """`count` tendons from a ring on base_link to a ring on tip_link, alternately twisted."""
        for k in range(count):
            ang = 2 * math.pi * (k + 0.5) / count
            sgn = 1.0 if k % 2 == 0 else -1.0
            a0 = ring_point(base_r * 1.15, ang - sgn * 0.15, base_z - 0.05)
            a1 = ring_point(base_r, ang - sgn * 0.15, base_z)
            b0 = ring_point(tip_r, ang + sgn * twist, tip_z)
            for pt in (a0, a1):
                pt[0] += shift[0]; pt[1] += shift[1]
            tendons.append({"name": "%s%d" % (name, k), "f_max": float(f_max * (1.0 + 0.5 * (k % 3 == 0))),
                            "via_points": [{"link": base_link, "pos": a0}, {"link": base_link, "pos": a1},
                                           {"link": tip_link, "pos": b0}]})

    add_ring("spine", 6, -1, 0.14, -0.04, torso, 0.11, 0.12, 120.0)
    add_ring("neck", 6, torso, 0.07, 0.36, head, 0.06, 0.05, 40.0)
    for side, (upper, elbow, roll, hand, sy) in arms.items():
        # 6 shoulder tendons torso -> upper arm, 3 elbow, 2 forearm roll, 2 wrist (13 per arm)
        add_ring(side + "_shoulder", 6, torso, 0.07, 0.40, upper, 0.045, -0.09, 60.0, shift=(0.0, sy * 0.20))
        for k in range(3):
            ang = (0.0, 2.4, -2.4)[k]
            tendons.append({"name": "%s_elbow%d" % (side, k), "f_max": 50.0,
                            "via_points": [{"link": upper, "pos": ring_point(0.035, ang, -0.10)},
                                           {"link": upper, "pos": ring_point(0.040, ang, -0.24)},
                                           {"link": elbow, "pos": ring_point(0.035, ang, -0.05)},
                                           {"link": elbow, "pos": ring_point(0.030, ang, -0.10)}]})
        for k in range(2):
            sgn = 1.0 if k == 0 else -1.0
            tendons.append({"name": "%s_roll%d" % (side, k), "f_max": 30.0,
                            "via_points": [{"link": elbow, "pos": ring_point(0.035, sgn * 0.9, -0.04)},
                                           {"link": elbow, "pos": ring_point(0.035, sgn * 0.9, -0.10)},
                                           {"link": roll, "pos": ring_point(0.030, -sgn * 0.9, -0.06)}]})
        for k in range(2):
            ang = (0.6, 3.7)[k]
            tendons.append({"name": "%s_wrist%d" % (side, k), "f_max": 25.0,
                            "via_points": [{"link": roll, "pos": ring_point(0.030, ang, -0.04)},
                                           {"link": roll, "pos": ring_point(0.030, ang, -0.10)},
                                           {"link": hand, "pos": ring_point(0.028, ang + 0.5, -0.04)}]})
    assert len(joints) == 20 and len(tendons) == 38
    return {"format": FORMAT_TAG, "name": "upper_body_synthetic", "gravity": [0.0, 0.0, -9.81],
            "joints": joints, "tendons": tendons,
            "muscle": {"kp": 10.0, "setpoint_scale": 0.03, "v_max": 8.0, "fl_width": 0.45, "kpe": 4.0,
                       "e0": 0.6, "fv_a": 0.25, "fv_n": 1.5}}


class UpperBodyRobot(RoboyRobot):

    _DIM_JOINT_ANGLE = 20
    _DIM_ACTION = 38
    _JOINT_ANGLE_SPACE = spaces.Box(low=-np.pi, high=np.pi, shape=(_DIM_JOINT_ANGLE,), dtype="float32")
    _JOINT_VEL_SPACE = spaces.Box(low=-np.pi / 6, high=np.pi / 6, shape=(_DIM_JOINT_ANGLE,), dtype="float32")
    _ACTION_SPACE = spaces.Box(low=-0.3, high=0.3, shape=(_DIM_ACTION,), dtype="float32")
    _DESCRIPTION = None

    @classmethod
    def get_description(cls) -> RobotDescription:
        if UpperBodyRobot._DESCRIPTION is None:
            UpperBodyRobot._DESCRIPTION = (RobotDescription.from_json(_DATA) if os.path.exists(_DATA)
                                           else RobotDescription(upper_body_spec()))
        return UpperBodyRobot._DESCRIPTION
