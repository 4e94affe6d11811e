"""``MsjRobot``: the 3-DOF / 8-tendon musculoskeletal-joint platform.

The three boxes are the reference's
(``/root/reference/gym_roboy/envs/robots/msj_robot.py:8-16``): joint angles
``+-pi``, joint velocities ``+-pi/6``, eight tendon set-points ``+-0.3``.

The physics description is build-defined (the reference has none): one ball
joint at the base origin written as three chained revolutes x-y-z, one moving
body (the top plate), eight tendons in four crossing pairs running motor
outlet -> guide ring (both on the base) -> top plate.  The numbers are chosen
so that

* the zero pose with zero set-points is an exact equilibrium
  (``test_simulation_client.py:14-19``, ``test_roboy_env.py:60-68``),
* one explicit integrator step of ``dt = 0.1`` (``ros_simulation_client.py:22``)
  is stable: geared tendon motors, i.e. large reflected inertia and friction,
* holding every set-point at its lower bound drives the platform into a joint
  limit (``test_simulation_client.py:54-68``): the +x tendon pairs are
  stronger than the -x pairs.
"""
import math

from ..._gymcompat import spaces
from .description import FORMAT_TAG, RobotDescription
from .roboy_robot import RoboyRobot


def msj_platform_spec() -> dict:
    """Build the ``roboy-tendon-robot/1`` description of the MSJ platform."""
    r_motor, z_motor = 0.120, -0.100   # motor outlets on the base
    r_guide, z_guide = 0.100, -0.020   # guide ring on the base
    r_top, z_top = 0.070, 0.090        # attachment ring on the top plate
    beta = math.radians(12.0)          # base angular offset inside a pair
    delta = math.radians(33.0)         # top angular offset inside a pair

    def ring(radius, angle, z):
        return [radius * math.cos(angle), radius * math.sin(angle), z]

    tendons = []
    for pair in range(4):
        theta = math.radians(45.0 + 90.0 * pair)
        strong = math.cos(theta) > 0.0  # pairs on the +x side
        for sign in (+1.0, -1.0):
            tendons.append({
                "name": "motor%d" % len(tendons),
                "f_max": 30.0 if strong else 6.0,
                "via_points": [
                    {"link": -1, "pos": ring(r_motor, theta - sign * beta, z_motor)},
                    {"link": -1, "pos": ring(r_guide, theta - sign * beta, z_guide)},
                    {"link": 2, "pos": ring(r_top, theta + sign * delta, z_top)},
                ],
            })

    tilt, yaw = 0.45, 0.60
    vmax = math.pi / 6
    def joint(name, parent, axis, limit, **kw):
        j = {"name": name, "parent": parent, "axis": axis, "origin": [0.0, 0.0, 0.0],
             "mass": 0.0, "com": [0.0, 0.0, 0.0], "inertia": [0.0] * 6,
             "armature": 0.20, "damping": 0.8, "limit": [-limit, limit],
             "max_velocity": vmax}
        j.update(kw)
        return j

    return {
        "format": FORMAT_TAG,
        "name": "msj_platform",
        "gravity": [0.0, 0.0, -9.81],
        "joints": [
            joint("sphere_axis0", -1, [1.0, 0.0, 0.0], tilt),
            joint("sphere_axis1", 0, [0.0, 1.0, 0.0], tilt),
            joint("sphere_axis2", 1, [0.0, 0.0, 1.0], yaw,
                  mass=0.25, com=[0.0, 0.0, 0.06],
                  inertia=[3.0e-4, 3.0e-4, 5.0e-4, 0.0, 0.0, 0.0]),
        ],
        "tendons": tendons,
        "muscle": {"kp": 10.0, "setpoint_scale": 0.1, "v_max": 8.0,
                   "fl_width": 0.45, "kpe": 4.0, "e0": 0.6,
                   "fv_a": 0.25, "fv_n": 1.5},
    }


class MsjRobot(RoboyRobot):
    """Boxes of the reference's MsjRobot (msj_robot.py:8-16); accessors live in the base class."""

    _DIM_JOINT_ANGLE, _DIM_ACTION = 3, 8
    _MAX_TENDON_LENGHT = 0.3    # sic: the reference's spelling, kept for drop-in
    _MAX_TENDON_VEL = 0.02      # unused by the reference as well (msj_robot.py:13)

    _ACTION_SPACE = spaces.Box(low=-_MAX_TENDON_LENGHT, high=_MAX_TENDON_LENGHT,
                               shape=(_DIM_ACTION,), dtype="float32")
    _JOINT_ANGLE_SPACE = spaces.Box(low=-math.pi, high=math.pi,
                                    shape=(_DIM_JOINT_ANGLE,), dtype="float32")
    _JOINT_VEL_SPACE = spaces.Box(low=-math.pi / 6, high=math.pi / 6,
                                  shape=(_DIM_JOINT_ANGLE,), dtype="float32")
    _DESCRIPTION = None

    @classmethod
    def get_description(cls) -> RobotDescription:
        if MsjRobot._DESCRIPTION is None:
            MsjRobot._DESCRIPTION = RobotDescription(msj_platform_spec())
        return MsjRobot._DESCRIPTION
