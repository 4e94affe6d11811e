"""Robot description: the data the batched physics step is evaluated on.

The reference hard-codes only the three ``Box`` bounds of ``MsjRobot``
(``/root/reference/gym_roboy/envs/robots/msj_robot.py:8-16``); the kinematic
and muscle model lives in the external CARDSflow simulator
(``/root/reference/README.md:34-41``) and is not available.  This module
defines the description format (``roboy-tendon-robot/1``) of the model spec in
``DESIGN.md`` §2:

* a tree of revolute joints; link ``i`` hangs off joint ``i`` (parent link
  ``parent[i]``, ``-1`` = the fixed base), its frame sits at the joint origin
  ``origin[i]`` (parent frame) and is rotated by ``q_i`` about ``axis[i]``;
* per link mass, centre of mass and inertia (about the COM, link frame),
  per joint armature (reflected actuator inertia), viscous damping, position
  limits (the *feasible* region) and a velocity limit;
* per tendon an ordered list of via-points ``(link, xyz in link frame)`` and
  the maximum isometric force of its Hill-type muscle;
* the muscle shape constants shared by all tendons.

Everything is plain data: the same description feeds the HIP library (as a
flat C struct, ``include/roboy_sim.h``) and, in the tests, the CPU oracle.
"""
import ctypes
import json

import numpy as np

FORMAT_TAG = "roboy-tendon-robot/1"

MUSCLE_DEFAULTS = {
    "kp": 10.0,             # activation per unit of normalised length error
    "setpoint_scale": 0.1,  # metres of tendon length per action-space unit
    "v_max": 8.0,           # max shortening speed, rest lengths per second
    "fl_width": 0.45,       # width of the Gaussian force-length curve
    "kpe": 4.0,             # passive element shape
    "e0": 0.6,              # passive strain at which f_PE = 1
    "fv_a": 0.25,           # Hill curvature of the shortening branch
    "fv_n": 1.5,            # eccentric force asymptote of the lengthening branch
}


class RobotDescriptionC(ctypes.Structure):
    """``rb_robot_desc`` of ``include/roboy_sim.h`` (field order must match)."""
    _fields_ = [
        ("n_q", ctypes.c_int32),
        ("n_t", ctypes.c_int32),
        ("n_vp", ctypes.c_int32),
        ("_pad", ctypes.c_int32),
        ("parent", ctypes.POINTER(ctypes.c_int32)),
        ("axis", ctypes.POINTER(ctypes.c_double)),
        ("origin", ctypes.POINTER(ctypes.c_double)),
        ("mass", ctypes.POINTER(ctypes.c_double)),
        ("com", ctypes.POINTER(ctypes.c_double)),
        ("inertia", ctypes.POINTER(ctypes.c_double)),
        ("armature", ctypes.POINTER(ctypes.c_double)),
        ("damping", ctypes.POINTER(ctypes.c_double)),
        ("q_lo", ctypes.POINTER(ctypes.c_double)),
        ("q_hi", ctypes.POINTER(ctypes.c_double)),
        ("qd_max", ctypes.POINTER(ctypes.c_double)),
        ("gravity", ctypes.c_double * 3),
        ("vp_offset", ctypes.POINTER(ctypes.c_int32)),
        ("vp_link", ctypes.POINTER(ctypes.c_int32)),
        ("vp_pos", ctypes.POINTER(ctypes.c_double)),
        ("f_max", ctypes.POINTER(ctypes.c_double)),
        ("kp", ctypes.c_double),
        ("setpoint_scale", ctypes.c_double),
        ("v_max", ctypes.c_double),
        ("fl_width", ctypes.c_double),
        ("kpe", ctypes.c_double),
        ("e0", ctypes.c_double),
        ("fv_a", ctypes.c_double),
        ("fv_n", ctypes.c_double),
    ]


def _arr(x, dtype, shape):
    a = np.ascontiguousarray(np.asarray(x, dtype=dtype))
    if a.shape != tuple(shape):
        raise ValueError("expected shape %s, got %s" % (tuple(shape), a.shape))
    return a


class RobotDescription:
    """Validated, flattened robot description."""

    def __init__(self, spec: dict):
        if spec.get("format") != FORMAT_TAG:
            raise ValueError("robot description: format must be %r" % FORMAT_TAG)
        self.name = str(spec.get("name", "robot"))
        joints = spec["joints"]
        tendons = spec["tendons"]
        n_q, n_t = len(joints), len(tendons)
        if n_q < 1 or n_t < 1:
            raise ValueError("robot description needs >=1 joint and >=1 tendon")
        self.n_q, self.n_t = n_q, n_t
        self.joint_names = [j.get("name", "joint%d" % i) for i, j in enumerate(joints)]
        self.tendon_names = [t.get("name", "tendon%d" % k) for k, t in enumerate(tendons)]

        self.parent = _arr([j["parent"] for j in joints], np.int32, (n_q,))
        for i, p in enumerate(self.parent):
            if not (-1 <= p < i):
                raise ValueError("joint %d: parent must be -1 or an earlier joint" % i)
        axis = _arr([j["axis"] for j in joints], np.float64, (n_q, 3))
        norm = np.linalg.norm(axis, axis=1)
        if np.any(norm < 1e-12):
            raise ValueError("joint axis must be non-zero")
        self.axis = np.ascontiguousarray(axis / norm[:, None])
        self.origin = _arr([j.get("origin", [0, 0, 0]) for j in joints], np.float64, (n_q, 3))
        self.mass = _arr([j.get("mass", 0.0) for j in joints], np.float64, (n_q,))
        self.com = _arr([j.get("com", [0, 0, 0]) for j in joints], np.float64, (n_q, 3))
        self.inertia = _arr([j.get("inertia", [0] * 6) for j in joints], np.float64, (n_q, 6))
        self.armature = _arr([j.get("armature", 0.0) for j in joints], np.float64, (n_q,))
        self.damping = _arr([j.get("damping", 0.0) for j in joints], np.float64, (n_q,))
        lim = _arr([j["limit"] for j in joints], np.float64, (n_q, 2))
        self.q_lo = np.ascontiguousarray(lim[:, 0])
        self.q_hi = np.ascontiguousarray(lim[:, 1])
        if np.any(self.q_lo >= 0) or np.any(self.q_hi <= 0):
            raise ValueError("joint limits must bracket the zero pose")
        self.qd_max = _arr([j["max_velocity"] for j in joints], np.float64, (n_q,))
        if np.any(self.mass < 0) or np.any(self.armature < 0) or np.any(self.damping < 0):
            raise ValueError("mass, armature and damping must be non-negative")
        self.gravity = _arr(spec.get("gravity", [0.0, 0.0, -9.81]), np.float64, (3,))

        offs, links, pos = [0], [], []
        for k, t in enumerate(tendons):
            vps = t["via_points"]
            if len(vps) < 2:
                raise ValueError("tendon %d needs at least two via-points" % k)
            for vp in vps:
                link = int(vp["link"])
                if not (-1 <= link < n_q):
                    raise ValueError("tendon %d: via-point link out of range" % k)
                links.append(link)
                pos.append(vp["pos"])
            offs.append(len(links))
        self.n_vp = len(links)
        self.vp_offset = _arr(offs, np.int32, (n_t + 1,))
        self.vp_link = _arr(links, np.int32, (self.n_vp,))
        self.vp_pos = _arr(pos, np.float64, (self.n_vp, 3))
        self.f_max = _arr([t["f_max"] for t in tendons], np.float64, (n_t,))
        if np.any(self.f_max <= 0):
            raise ValueError("f_max must be positive")

        muscle = dict(MUSCLE_DEFAULTS)
        muscle.update(spec.get("muscle", {}))
        unknown = set(muscle) - set(MUSCLE_DEFAULTS)
        if unknown:
            raise ValueError("unknown muscle parameters: %s" % sorted(unknown))
        self.muscle = {k: float(v) for k, v in muscle.items()}
        self._spec = spec
        self._c_struct = None

    # ------------------------------------------------------------------
    @classmethod
    def from_json(cls, path):
        with open(path) as fh:
            return cls(json.load(fh))

    def to_dict(self):
        return json.loads(json.dumps(self._spec))

    def to_json(self, path):
        with open(path, "w") as fh:
            json.dump(self._spec, fh, indent=1)
            fh.write("\n")

    def as_c_struct(self) -> RobotDescriptionC:
        """Flat ``rb_robot_desc`` whose pointers alias this object's arrays
        (kept alive by ``self``)."""
        if self._c_struct is None:
            c = RobotDescriptionC()
            c.n_q, c.n_t, c.n_vp = self.n_q, self.n_t, self.n_vp
            i32 = ctypes.POINTER(ctypes.c_int32)
            f64 = ctypes.POINTER(ctypes.c_double)
            for name in ("parent", "vp_offset", "vp_link"):
                setattr(c, name, getattr(self, name).ctypes.data_as(i32))
            for name in ("axis", "origin", "mass", "com", "inertia", "armature",
                         "damping", "q_lo", "q_hi", "qd_max", "vp_pos", "f_max"):
                setattr(c, name, getattr(self, name).ctypes.data_as(f64))
            c.gravity[:] = self.gravity.tolist()
            for name, value in self.muscle.items():
                setattr(c, name, value)
            self._c_struct = c
        return self._c_struct
