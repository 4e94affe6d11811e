"""rocprofv3 target: two PPO iterations with the fused kernels at 262 144 envs (which kernels the update still spends time in)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gym_roboy_amd.envs.robots import MsjRobot
from gym_roboy_amd.envs.vec_env import RoboyVecEnv
from gym_roboy_amd.ppo import PPO
env = RoboyVecEnv(MsjRobot(), 262144)
agent = PPO(env, ent_coef=0.1, device="cuda", reward_scale=0.01, use_graphs=True, fused_policy=True, fused_update=True)
for _ in range(3):
    agent.update(agent.collect())
torch.cuda.synchronize()
