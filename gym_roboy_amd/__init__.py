"""gym-roboy on MI355X: batched tendon-robot physics behind gym-roboy's
``SimulationClient`` / ``RoboyRobot`` / ``RoboyEnv`` plugin surface.

Registers ``msj-control-v0`` (the id the reference registers,
``/root/reference/gym_roboy/__init__.py:3-6``) and ``msj-control-v1`` (the id
its README and BASELINE.json use).  ``make`` is ``gym.make`` when the real gym
is installed, the in-repo registry otherwise.
"""
from . import _gymcompat

for _env_id in ("msj-control-v0", "msj-control-v1"):
    _gymcompat.register(id=_env_id, entry_point="gym_roboy_amd.envs:RoboyEnv")
    if _gymcompat.HAVE_REAL_GYM:  # pragma: no cover
        from gym.envs.registration import register as _gym_register
        try:
            _gym_register(id=_env_id, entry_point="gym_roboy_amd.envs:RoboyEnv")
        except Exception:
            pass

make = _gymcompat.make
__version__ = "0.1.0"
