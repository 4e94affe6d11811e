"""Throughput of the fused env layer (RoboyVecEnv.step_dev) at a few batch sizes (not the headline metric)."""
import os, sys, time
sys.path.insert(0, "/root/repo")
import torch
from gym_roboy_amd.envs.robots import MsjRobot, UpperBodyRobot
from gym_roboy_amd.envs.vec_env import RoboyVecEnv
ROBOT = UpperBodyRobot() if os.environ.get("VECENV_ROBOT", "msj") == "upper" else MsjRobot()
INTEG = os.environ.get("VECENV_INTEGRATOR", "euler")
SIZES = [int(x) for x in os.environ.get("VECENV_SIZES", "4096,65536,262144,2097152").split(",")]
# VECENV_DESYNC=1: the envs' episode counters start uniformly spread over an episode, as they are in training once goals have been
# reached here and there - every step then ends one episode in 400 (a launch waits for the waves that redraw a goal); default: all
# episodes in lock-step (they end together every 400 steps)
DESYNC = os.environ.get("VECENV_DESYNC", "0") == "1"
# VECENV_KERNEL: rb_select_kernel on the env's handle (0 / unset: the library's choice; 1: one env per lane; 5: two lanes per env)
KERNEL = int(os.environ.get("VECENV_KERNEL", "0"))
# VECENV_GRAPH=1: the steps as replays of a captured 50-step graph (small batches: a Python launch loop is slower than the kernels)
GRAPH = os.environ.get("VECENV_GRAPH", "0") == "1"
for n in SIZES:
    with torch.cuda.stream(torch.cuda.Stream()):
        env = RoboyVecEnv(ROBOT, n, integrator=INTEG)
        st = torch.cuda.current_stream(); env.sim.set_stream(st.cuda_stream)
        if KERNEL:
            env.sim.select_kernel(KERNEL)
        if DESYNC:
            import numpy as np
            rng = np.random.default_rng(3)
            lo, hi = ROBOT.get_joint_angles_space().low, ROBOT.get_joint_angles_space().high
            env.reset()
            env.set_goal(rng.uniform(lo, hi, (n, env.n_q)).astype(np.float32), rng.integers(1, 401, n).astype(np.uint32))
        acts = [torch.rand((n, env.n_t), device="cuda") * 2 - 1 for _ in range(4)]
        obs = torch.empty((n, 3 * env.n_q), device="cuda"); rew = torch.empty(n, device="cuda"); done = torch.empty(n, dtype=torch.int32, device="cuda")
        steps = 2000 if n <= 65536 else 300
        for t in range(50): env.step_dev(acts[t % 4].data_ptr(), obs.data_ptr(), rew.data_ptr(), done.data_ptr())
        torch.cuda.synchronize()
        if GRAPH:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=st):
                for t in range(50): env.step_dev(acts[t % 4].data_ptr(), obs.data_ptr(), rew.data_ptr(), done.data_ptr())
            g.replay(); torch.cuda.synchronize()
            steps = (steps // 50) * 50
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter(); e0.record(st)
        if GRAPH:
            for t in range(steps // 50): g.replay()
        else:
            for t in range(steps): env.step_dev(acts[t % 4].data_ptr(), obs.data_ptr(), rew.data_ptr(), done.data_ptr())
        e1.record(st); torch.cuda.synchronize(); wall = time.perf_counter() - t0
        us = e0.elapsed_time(e1) * 1e3 / steps
        print(INTEG, "desync" if DESYNC else "lockstep", "kernel %d (%s)" % (env.sim.info()["kernel"], "graph" if GRAPH else "eager"), "fused env step n=%d: %.2f us/step (events), %.3e env-steps/s wall, %.1f GB/s algorithmic" % (n, us, n * steps / wall, n * (4 * (4 * env.n_q + env.n_t + 1) + 4 * (4 * env.n_q + 6)) / us / 1e3))
        env.close()
