#!/bin/bash
# round 4: sub-range entry points + PPO rollout in two chains: tests, then rollout time per step with one / two chains
cd /root/repo
mkdir -p gpurun_out/r4_a
timeout -k 10 900 python -m pytest tests/test_ppo.py tests/test_physics_gpu.py tests/test_env_layer_gpu.py tests/test_env_golden_gpu.py tests/test_tree_robot_gpu.py -x -q -m gpu > gpurun_out/r4_a/chains_ppo_tests.log 2>&1; rc=$?
tail -15 gpurun_out/r4_a/chains_ppo_tests.log
[ $rc -ne 0 ] && exit $rc
python - <<'PY' 2>&1 | tee gpurun_out/r4_a/ppo_chains.log
import time, torch
from gym_roboy_amd.envs.robots import MsjRobot, UpperBodyRobot
from gym_roboy_amd.envs.vec_env import RoboyVecEnv
from gym_roboy_amd.ppo import PPO
for robot_name, n in (("msj", 16384), ("msj", 32768), ("msj", 65536), ("msj", 262144), ("msj", 1048576), ("upper", 32768), ("upper", 65536)):
    for chains in (1, 2):
        with torch.cuda.stream(torch.cuda.Stream()):
            env = RoboyVecEnv(MsjRobot() if robot_name == "msj" else UpperBodyRobot(), n)
            if robot_name == "upper":
                env.sim.select_kernel(1)
            agent = PPO(env, ent_coef=0.1, device="cuda", reward_scale=0.01, use_graphs=True, rollout_chains=chains, n_steps=64 if n > 500000 else 128)
            agent.collect(); agent.collect(); torch.cuda.synchronize()
            reps = 5
            t0 = time.perf_counter()
            for _ in range(reps): agent.collect()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / reps
            print("%s %8d envs, %d chain(s): rollout %.3f ms = %.2f us per step" % (robot_name, n, agent.rollout_chains, dt * 1e3, dt * 1e6 / agent.n_steps), flush=True)
            env.close()
PY
