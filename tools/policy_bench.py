#!/usr/bin/env python3
"""Policy step of the PPO consumer on one MI355X: the fused MFMA kernel (csrc/mlp_policy.hip) against the torch MlpPolicy.act
it replaces, per batch size; then a PPO iteration with and without it (tools/ppo_profile.py has the long form)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gym_roboy_amd.ppo import FusedPolicyStep, MlpPolicy, PPO
from gym_roboy_amd.envs.robots import MsjRobot
from gym_roboy_amd.envs.vec_env import RoboyVecEnv


def timed(fn, reps):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


obs_dim, act_dim = 9, 8
policy = MlpPolicy(obs_dim, act_dim).cuda()
fused = FusedPolicyStep(policy)
flops = 2 * 2 * (obs_dim * 64 + 64 * 64) + 2 * 64 * (act_dim + 1)          # useful multiply-adds x 2
for n in [int(a) for a in sys.argv[1:]] or [4096, 65536, 262144, 2097152]:
    obs = torch.rand(n, obs_dim, device="cuda") * 2 - 1
    act, logp, val = torch.empty(n, act_dim, device="cuda"), torch.empty(n, device="cuda"), torch.empty(n, device="cuda")
    packed = fused.pack()
    t_f = timed(lambda: fused.act_into(obs, act, logp, val, packed=packed), 50)
    t_p = timed(lambda: fused.pack(), 50)
    t_t = timed(lambda: policy.act(obs), 20)
    print("n=%8d: fused kernel %8.1f us (%.1f useful TFLOP/s; weight gather %.1f us once per rollout), torch MlpPolicy.act %8.1f us (eager)"
          % (n, t_f, flops * n / t_f / 1e6, t_p, t_t), flush=True)

from gym_roboy_amd.ppo import FusedPolicyGrad
fg = FusedPolicyGrad(policy)
for B in (2097152, 8388608):
    mb = [torch.rand(B, obs_dim, device="cuda"), torch.randn(B, act_dim, device="cuda") * 0.5, torch.randn(B, device="cuda"),
          torch.randn(B, device="cuda") - 8.0, torch.randn(B, device="cuda"), torch.randn(B, device="cuda")]
    t_g = timed(lambda: fg.run(*mb, 0.2, 0.5, 0.1), 5)
    perm = torch.randperm(B, device="cuda")
    t_i = timed(lambda: [t[perm].contiguous() for t in mb], 5)
    flops_fb = 3 * flops                                   # forward + two backward products per weight
    print("minibatch of %d samples: fused gradient (3 launches) %.2f ms = %.1f useful TFLOP/s; torch gather of the minibatch tensors %.2f ms"
          % (B, t_g / 1e3, flops_fb * B / t_g / 1e6, t_i / 1e3), flush=True)
    del mb, perm

for n in (65536, 262144):
    for fused in (False, True):
        env = RoboyVecEnv(MsjRobot(), n)
        agent = PPO(env, ent_coef=0.1, device="cuda", reward_scale=0.01, use_graphs=True, fused_policy=fused, fused_update=fused)
        roll = agent.collect(); agent.update(roll); torch.cuda.synchronize()
        tc = tu = 0.0
        iters = 3
        for _ in range(iters):
            t0 = time.perf_counter(); roll = agent.collect(); torch.cuda.synchronize(); t1 = time.perf_counter()
            agent.update(roll); torch.cuda.synchronize(); t2 = time.perf_counter()
            tc += t1 - t0; tu += t2 - t1
        print("PPO iteration, N=%d, graphs, %s: rollout %.1f ms (%.1f us per vectorised step), update %.1f ms; %.3g timesteps/s end to end"
              % (n, "fused policy step + fused gradient" if fused else "torch policy + autograd", tc / iters * 1e3,
                 tc / iters / agent.n_steps * 1e6, tu / iters * 1e3, iters * agent.n_steps * n / (tc + tu)), flush=True)
        env.close()
