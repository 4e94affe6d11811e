"""Where a PPO iteration spends its time on one MI355X: rollout collection vs update."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gym_roboy_amd.envs.robots import MsjRobot
from gym_roboy_amd.envs.vec_env import RoboyVecEnv
from gym_roboy_amd.ppo import PPO

import itertools
for n, graphs in itertools.product([int(a) for a in sys.argv[1:]] or [4096, 65536], (False, True)):
    env = RoboyVecEnv(MsjRobot(), n)
    agent = PPO(env, ent_coef=0.1, device="cuda", reward_scale=0.01, use_graphs=graphs)
    roll = agent.collect(); agent.update(roll)
    torch.cuda.synchronize()
    tc = tu = 0.0
    iters = 5
    for _ in range(iters):
        t0 = time.perf_counter(); roll = agent.collect(); torch.cuda.synchronize(); t1 = time.perf_counter()
        agent.update(roll); torch.cuda.synchronize(); t2 = time.perf_counter()
        tc += t1 - t0; tu += t2 - t1
    steps = iters * agent.n_steps * n
    print("N=%d %s: collect %.1f ms (%.1f us per vectorised step), update %.1f ms per iteration; %.3g timesteps/s end to end"
          % (n, "HIP graphs" if graphs else "eager     ", tc / iters * 1e3, tc / iters / agent.n_steps * 1e6, tu / iters * 1e3, steps / (tc + tu)), flush=True)
    env.close()
