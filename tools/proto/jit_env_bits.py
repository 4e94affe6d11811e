"""One-off check (round 4): are the hiprtc-built plain step and fused env step of a random robot bit-identical in the one-wave (1) and
split (4) forms now that the generated text writes its fused multiply-adds out?"""
import sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
from test_random_robots_gpu import random_tree_robot
from host_env_model import HipStepper, HostEnvModel
from gym_roboy_amd.envs.vec_env import RoboyVecEnv
for seed in (4, 9, 10):
    for kernel in (1, 4):
        robot, desc = random_tree_robot(seed)
        n, env_seed, max_len = 130, 5, 7
        try:
            vec = RoboyVecEnv(robot, n, seed=env_seed, auto_reset=True, max_episode_length=max_len, joint_vel_penalty=True)
            vec.sim.select_kernel(kernel)
        except Exception as exc:
            print("robot", seed, "kernel", kernel, "not available:", str(exc)[:80]); continue
        stepper = HipStepper(robot, n, env_seed); stepper.sim.select_kernel(kernel)
        host = HostEnvModel(robot, stepper, n, env_seed, max_len, True, True, True)
        vec.reset(); host.goal = host.draw(np.ones(n, bool))
        rng = np.random.default_rng(2)
        worst = 0.0
        for t in range(12):
            a = rng.uniform(-1, 1, (n, desc.n_t)).astype(np.float32)
            obs, rew, done, _ = vec.step(a)
            h_obs, h_rew, h_done, margin = host.step(a)
            worst = max(worst, float(np.abs(obs - h_obs).max()))
        print("robot", seed, "kernel", kernel, "max |obs - replay| over 12 steps:", worst)
        vec.close(); stepper.close()
