"""Minimal gym-compatible surface used when the real ``gym`` package is absent.

The reference depends on ``gym`` for exactly three things
(``/root/reference/gym_roboy/envs/roboy_env.py:3-4,10``,
``/root/reference/gym_roboy/__init__.py:1``): ``spaces.Box``, the ``GoalEnv``
base class and the ``register``/``make`` registry.  Neither ``gym`` nor
``gymnasium`` is installed in this image or on the GPU box, so the package
carries this small stand-in and prefers the real package when it imports.
"""
import importlib

import numpy as np

try:  # pragma: no cover - real gym is not installed in this image
    import gym as _real_gym
    from gym import spaces as _real_spaces
    HAVE_REAL_GYM = True
except Exception:  # ModuleNotFoundError here
    _real_gym = None
    _real_spaces = None
    HAVE_REAL_GYM = False


class Box:
    """Axis-aligned bounded box in R^n (``gym.spaces.Box`` subset)."""

    def __init__(self, low, high, shape=None, dtype="float32"):
        self.dtype = np.dtype(dtype)
        if shape is None:
            low = np.asarray(low)
            high = np.asarray(high)
            if low.shape != high.shape:
                raise ValueError("Box: low and high must have the same shape")
            self.shape = tuple(low.shape)
            self.low = low.astype(self.dtype)
            self.high = high.astype(self.dtype)
        else:
            self.shape = tuple(shape)
            self.low = np.full(self.shape, low, dtype=self.dtype)
            self.high = np.full(self.shape, high, dtype=self.dtype)

    def sample(self):
        # draws from numpy's global generator so ``env.seed`` (which seeds the
        # global generator, roboy_env.py:114-115) makes sampling reproducible
        return np.random.uniform(low=self.low, high=self.high,
                                 size=self.shape).astype(self.dtype)

    def contains(self, x):
        x = np.asarray(x)
        return bool(x.shape == self.shape
                    and np.all(x >= self.low) and np.all(x <= self.high))

    def __contains__(self, x):
        return self.contains(x)

    def __repr__(self):
        return "Box(%s, %s)" % (self.low, self.high)

    def __eq__(self, other):
        return (isinstance(other, Box) and self.shape == other.shape
                and np.array_equal(self.low, other.low)
                and np.array_equal(self.high, other.high))


class Env:
    metadata = {"render.modes": []}
    reward_range = (-float("inf"), float("inf"))
    action_space = None
    observation_space = None

    def step(self, action):
        raise NotImplementedError

    def reset(self):
        raise NotImplementedError

    def render(self, mode="human"):
        raise NotImplementedError

    def close(self):
        pass

    def seed(self, seed=None):
        return None


class GoalEnv(Env):
    """Marker base class, as ``gym.GoalEnv`` is for the reference env."""


class _Spaces:
    Box = Box


_REGISTRY = {}


def register(id, entry_point, kwargs=None, **_ignored):
    _REGISTRY[id] = (entry_point, dict(kwargs or {}))


def make(id, **kwargs):
    if id not in _REGISTRY:
        raise KeyError("no registered env with id %r (known: %s)"
                       % (id, sorted(_REGISTRY)))
    entry_point, defaults = _REGISTRY[id]
    if isinstance(entry_point, str):
        mod_name, attr = entry_point.split(":")
        entry_point = getattr(importlib.import_module(mod_name), attr)
    merged = dict(defaults)
    merged.update(kwargs)
    return entry_point(**merged)


if HAVE_REAL_GYM:  # pragma: no cover
    spaces = _real_spaces
    BaseGoalEnv = getattr(_real_gym, "GoalEnv", _real_gym.Env)
else:
    spaces = _Spaces
    BaseGoalEnv = GoalEnv
