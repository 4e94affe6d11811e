"""Environment layer: the single-env ``RoboyEnv`` (the reference's class) and the
GPU-batched ``RoboyVecEnv`` (N envs advanced by one fused kernel per step)."""
from .roboy_env import RoboyEnv
from .vec_env import RoboyVecEnv

__all__ = ["RoboyEnv", "RoboyVecEnv"]
