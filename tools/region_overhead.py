#!/usr/bin/env python3
"""What a short timed region of bench.py costs beyond its K kernel launches (one GPU): wall time of
synchronize -> K per-step launches [-> statistics reduction] -> synchronize, for graph replay and direct launches."""
import ctypes
import statistics
import sys
import time

import torch

sys.path.insert(0, ".")
from gym_roboy_amd import _native as nat
from gym_roboy_amd.envs.robots import MsjRobot
from gym_roboy_amd.envs.simulations.hip_simulation_client import HipBatchSimulation

n = 262144
torch.cuda.set_device(0)
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    sim = HipBatchSimulation(MsjRobot(), n, integrator="rk4")
    sim.set_stream(s.cuda_stream)
    RING = 4
    ring = torch.empty(RING * n * 8, dtype=torch.float32, device="cuda")
    for r in range(RING):
        sim.fill_actions_dev(ring.data_ptr() + 4 * r * n * 8, r)
    buf = torch.zeros(8, dtype=torch.float64, device="cuda")

    def region(k, graph, stats):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if k:
            sim.rollout_dev(ring.data_ptr(), RING, k, 0.3, use_graph=graph)
        if stats:
            nat.check(sim._lib.rb_env_stats_dev(sim.handle, ctypes.c_void_p(buf.data_ptr()), 0))
        torch.cuda.synchronize()
        return time.perf_counter() - t0

    for k in (0, 1, 20, 100):
        for graph in (True, False):
            for stats in (False, True):
                for _ in range(5):
                    region(k, graph, stats)
                w = statistics.median(region(k, graph, stats) for _ in range(41))
                print("K=%3d %-6s stats=%d: %7.1f us per region%s" % (
                    k, "graph" if graph else "direct", stats, w * 1e6,
                    "  (%.2f us per step)" % (w * 1e6 / k) if k else ""), flush=True)
