"""Probe: can a whole PPO rollout (policy forward + sampling + fused env step, T steps) be captured
in one torch CUDA graph with the env kernel launched through ctypes?  Timing vs the eager loop."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gym_roboy_amd.envs.robots import MsjRobot
from gym_roboy_amd.envs.vec_env import RoboyVecEnv
from gym_roboy_amd.ppo import MlpPolicy

N, T = int(os.environ.get("PROBE_N", "4096")), 128
env = RoboyVecEnv(MsjRobot(), N)
policy = MlpPolicy(9, 8).cuda()
side = torch.cuda.Stream()
obs_buf = torch.zeros((T + 1, N, 9), device="cuda")
rew = torch.zeros((T, N), device="cuda")
done = torch.zeros((T, N), dtype=torch.int32, device="cuda")
acts = torch.zeros((T, N, 8), device="cuda")
logps = torch.zeros((T, N), device="cuda")
vals = torch.zeros((T, N), device="cuda")
obs_buf[0].copy_(torch.as_tensor(env.reset(), device="cuda"))


def rollout():
    for t in range(T):
        a, logp, v = policy.act(obs_buf[t])
        acts[t].copy_(a); logps[t].copy_(logp); vals[t].copy_(v)
        clipped = a.clamp(-1.0, 1.0).contiguous()
        env.step_dev(clipped.data_ptr(), obs_buf[t + 1].data_ptr(), rew[t].data_ptr(), done[t].data_ptr())
    obs_buf[0].copy_(obs_buf[T])


with torch.cuda.stream(side):
    env.sim.set_stream(side.cuda_stream)
    for _ in range(2):
        rollout()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        rollout()
    torch.cuda.synchronize()
    eager = (time.perf_counter() - t0) / 3
print("eager rollout of %d steps x %d envs: %.2f ms (%.1f us per step)" % (T, N, eager * 1e3, eager / T * 1e6), flush=True)

g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=side):
    rollout()
torch.cuda.synchronize()
before = rew.sum().item()
for _ in range(2):
    g.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    g.replay()
torch.cuda.synchronize()
graph = (time.perf_counter() - t0) / 10
print("graph replay: %.2f ms (%.1f us per step); reward sums differ between replays: %s; finite: %s"
      % (graph * 1e3, graph / T * 1e6, before != rew.sum().item(), bool(torch.isfinite(obs_buf).all())), flush=True)
a0 = acts.clone(); g.replay(); torch.cuda.synchronize()
print("actions differ between replays (RNG advances):", bool((a0 != acts).any()))
print(env.stats())
env.close()
