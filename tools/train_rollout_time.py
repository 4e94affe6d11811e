#!/usr/bin/env python3
"""Rollout time per step of the training driver's PPO at world size 1 and 2 (two ranks sharing the one GPU, gloo):
the captured rollout graph is replayed at every world size (gym_roboy_amd/train_parallel.py).  Launch:

    python tools/train_rollout_time.py                      # world 1
    python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29701 tools/train_rollout_time.py
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gym_roboy_amd.envs.robots import MsjRobot
from gym_roboy_amd.envs.vec_env import RoboyVecEnv
from gym_roboy_amd.ppo import PPO

world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
dist = None
torch.cuda.set_device(0)
if world > 1:
    import torch.distributed as dist
    dist.init_process_group("gloo")
n_envs = int(os.environ.get("ROLLOUT_ENVS", "4096"))
env = RoboyVecEnv(MsjRobot(), n_envs, seed=0, env_id_offset=rank * n_envs)
agent = PPO(env, n_steps=128, ent_coef=0.1, device="cuda", dist=dist, reward_scale=0.01, use_graphs=True)
agent.collect(); agent.collect()
torch.cuda.synchronize()
# The ranks of this rehearsal share ONE GPU, and two processes on one GPU are time-sliced by the driver (measured: both
# measuring at once 85 us per step against 19 alone).  What is to be shown is that a rank of a multi-rank run replays
# its rollout graph at the single-rank cost, so the ranks measure in turns: one runs, the others wait at a barrier.
iters = 20
dt = None
for turn in range(world):
    if dist is not None:
        dist.barrier()
    if turn == rank:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            agent.collect()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
if dist is not None:
    dist.barrier()
out = {"world": world, "rank": rank, "envs_per_rank": n_envs, "graphs": agent.use_graphs and agent._rollout_graph is not None,
       "rollout_us_per_step": dt / (iters * 128) * 1e6}
if dist is not None:
    t = torch.tensor([out["rollout_us_per_step"]], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    out["rollout_us_per_step_max_over_ranks"] = float(t.item())
if rank == 0:
    print(json.dumps(out))
if dist is not None:
    dist.destroy_process_group()
