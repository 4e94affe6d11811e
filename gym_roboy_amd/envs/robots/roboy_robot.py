"""Robot value types: ``RobotState`` and the ``RoboyRobot`` base class.

Mirrors the public surface of
``/root/reference/gym_roboy/envs/robots/roboy_robot.py:6-95`` (same class,
method and keyword names, same semantics) so the reference's own unit tests
(``gym_roboy/envs/tests/test_robot_state.py``) read the same against this
package.  ``typeguard`` is not installed here; the one check the reference
relies on it for (``is_feasible`` must be a Python ``bool``,
``roboy_robot.py:7-8``) is done by hand.

On top of the reference surface a robot also carries a *description*: the
kinematic tree, inertias, tendon via-points and muscle parameters the HIP
physics step needs (the reference has none; its physics is the external
CARDSflow simulator).  See ``description.py``.
"""
import numpy as np

from ..._gymcompat import spaces


def _require_bool(value, name):
    if not isinstance(value, (bool, np.bool_)):
        raise TypeError('type of argument "%s" must be bool; got %s instead'
                        % (name, type(value).__name__))
    return bool(value)


class RobotState:
    """Joint angles, joint velocities and the feasibility flag of one robot."""

    def __init__(self, joint_angles, joint_vels, is_feasible):
        self.joint_angles = np.array(joint_angles)
        self.joint_vels = np.array(joint_vels)
        self.is_feasible = _require_bool(is_feasible, "is_feasible")

    @classmethod
    def interpolate(cls, state1, state2):
        """Midpoint of two states; feasible only if both are
        (reference ``roboy_robot.py:12-18``)."""
        if not (isinstance(state1, cls) and isinstance(state2, cls)):
            raise AssertionError("interpolate needs two RobotState objects")
        return cls(joint_angles=0.5 * (state1.joint_angles + state2.joint_angles),
                   joint_vels=0.5 * (state1.joint_vels + state2.joint_vels),
                   is_feasible=state1.is_feasible and state2.is_feasible)

    def __repr__(self):
        return "RobotState(q=%s, qd=%s, feasible=%s)" % (
            self.joint_angles, self.joint_vels, self.is_feasible)


class RoboyRobot:
    """Abstract robot: three spaces, state factories and normalisation."""

    # A concrete robot declares its three boxes as class attributes; the
    # accessors below are the reference's interface (roboy_robot.py:24-33).
    _ACTION_SPACE = None
    _JOINT_ANGLE_SPACE = None
    _JOINT_VEL_SPACE = None

    @classmethod
    def _declared(cls, attribute) -> spaces.Box:
        box = getattr(cls, attribute)
        if box is None:
            raise NotImplementedError("%s does not declare %s" % (cls.__name__, attribute))
        return box

    @classmethod
    def get_action_space(cls) -> spaces.Box:
        return cls._declared("_ACTION_SPACE")

    @classmethod
    def get_joint_angles_space(cls) -> spaces.Box:
        return cls._declared("_JOINT_ANGLE_SPACE")

    @classmethod
    def get_joint_vels_space(cls) -> spaces.Box:
        return cls._declared("_JOINT_VEL_SPACE")

    @classmethod
    def get_description(cls):
        """The physics description (``description.RobotDescription``)."""
        raise NotImplementedError

    # --- state factories -------------------------------------------------
    @classmethod
    def _zeros(cls, space):
        return np.zeros(space.shape)

    @classmethod
    def new_random_state(cls) -> RobotState:
        # The reference draws the *velocities* from the angle space too
        # (roboy_robot.py:38); kept, because its tests only need "random".
        angle_space = cls.get_joint_angles_space()
        return RobotState(angle_space.sample(), angle_space.sample(), True)

    @classmethod
    def new_zero_state(cls) -> RobotState:
        return RobotState(cls._zeros(cls.get_joint_angles_space()),
                          cls._zeros(cls.get_joint_vels_space()), True)

    @classmethod
    def new_random_zero_vels_state(cls) -> RobotState:
        return RobotState(cls.get_joint_angles_space().sample(),
                          cls._zeros(cls.get_joint_vels_space()), True)

    @classmethod
    def new_random_zero_angles_state(cls) -> RobotState:
        return RobotState(cls._zeros(cls.get_joint_angles_space()),
                          cls.get_joint_vels_space().sample(), True)

    @classmethod
    def new_max_state(cls) -> RobotState:
        return RobotState(cls.get_joint_angles_space().high,
                          cls.get_joint_vels_space().high, False)

    @classmethod
    def new_min_state(cls) -> RobotState:
        return RobotState(cls.get_joint_angles_space().low,
                          cls.get_joint_vels_space().low, False)

    @classmethod
    def new_state(cls, joint_angle, joint_vel, is_feasible) -> RobotState:
        """Wrap raw simulator output; the angles must lie inside the angle box
        (reference ``roboy_robot.py:71-78``; the velocity check is disabled
        there too)."""
        is_feasible = _require_bool(is_feasible, "is_feasible")
        joint_angle = np.asarray(joint_angle)
        joint_vel = np.asarray(joint_vel)
        assert cls.get_joint_angles_space().contains(joint_angle), joint_angle
        return RobotState(joint_angle, joint_vel, is_feasible)

    # --- normalisation ---------------------------------------------------
    def normalize_state(self, state: RobotState) -> RobotState:
        angles = self.get_joint_angles_space()
        vels = self.get_joint_vels_space()
        return RobotState(
            joint_angles=self._normalize_between_minus1_and1(
                state.joint_angles, angles.high, angles.low),
            joint_vels=self._normalize_between_minus1_and1(
                state.joint_vels, vels.high, vels.low),
            is_feasible=state.is_feasible)

    @staticmethod
    def _normalize_between_minus1_and1(val, max_val, min_val):
        """Affine map [min, max] -> [-1, 1] (reference ``roboy_robot.py:93-95``)."""
        return (2 * val - max_val - min_val) / (max_val - min_val)
