from .roboy_env import RoboyEnv
