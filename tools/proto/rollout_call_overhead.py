#!/usr/bin/env python3
"""Probe (round 6): what a rb_rollout_dev CALL costs beyond its steps, without a profiler in the way (rocprofv3's
kernel tracing serialises the two chains: profiles/r6_a/call_timeline_under_rocprof.log).  For K in a list of step counts the wall
time of  synchronize -> rb_rollout_dev(K) -> synchronize  and the HIP-event time around the call are measured (median of many calls);
a straight line  time(K) = per_call + per_step * K  is fitted: the intercept is what the driver's K = 20 pays per call.  Decomposition:

    floor       the same bracket around ONE launch of a 2-us kernel (a 64-env handle): launch + completion + host wake-up of this stack
    one chain   rollout as one launch per step (graphs, no fork / join)
    two chains  the shipped form (fork skipped on an idle stream, eager head, two graphs, join)

usage: python3 tools/proto/rollout_call_overhead.py  [reps]"""
import os
import statistics
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from gym_roboy_amd.envs.robots import MsjRobot
from gym_roboy_amd.envs.simulations import HipBatchSimulation

REPS = int(sys.argv[1]) if len(sys.argv) > 1 else 60
KS = (8, 12, 20, 40, 100, 400)
n = 262144
st = torch.cuda.Stream()


def bracket(fn, reps):
    walls, evs = [], []
    for _ in range(reps):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record(st)
        fn()
        e1.record(st)
        torch.cuda.synchronize()
        walls.append((time.perf_counter() - t0) * 1e6)
        evs.append(e0.elapsed_time(e1) * 1e3)
    return statistics.median(walls), statistics.median(evs)


with torch.cuda.stream(st):
    small = HipBatchSimulation(MsjRobot(), 64, integrator="euler")
    small.set_stream(st.cuda_stream)
    a64 = torch.zeros(64 * 8, device="cuda")
    for _ in range(20):
        small.step_dev(a64.data_ptr(), 0.3)
    w, e = bracket(lambda: small.step_dev(a64.data_ptr(), 0.3), 200)
    print("floor: synchronize -> one launch of a 64-env step -> synchronize: wall %.1f us, events %.1f us" % (w, e))
    w0, e0 = bracket(lambda: None, 200)
    print("       ... -> nothing -> ...: wall %.1f us, events %.1f us (the bracket itself: two event records + the synchronisation)" % (w0, e0))
    for chains in (1, 2):
        sim = HipBatchSimulation(MsjRobot(), n, integrator="rk4", seed=1)
        sim.set_stream(st.cuda_stream)
        sim.set_rollout_chains(chains)
        ring = torch.empty(4 * n * 8, dtype=torch.float32, device="cuda")
        for r in range(4):
            sim.fill_actions_dev(ring.data_ptr() + 4 * r * n * 8, r)
        rows = []
        for k in KS:
            for _ in range(3):
                sim.rollout_dev(ring.data_ptr(), 4, k, 0.3, use_graph=True)
            w, e = bracket(lambda: sim.rollout_dev(ring.data_ptr(), 4, k, 0.3, use_graph=True), REPS if k <= 100 else max(8, REPS // 6))
            rows.append((k, w, e))
            print("chains %d  K %4d: wall %8.1f us (%.2f per step), events %8.1f us (%.2f per step)" % (chains, k, w, w / k, e, e / k))
        ks = np.array([r[0] for r in rows], float)
        for name, col in (("wall", 1), ("events", 2)):
            y = np.array([r[col] for r in rows])
            b, a = np.polyfit(ks, y, 1)
            print("chains %d  %-6s fit: %.1f us per call + %.3f us per step   (K = 20: %.2f us per step = %.2f + %.2f)" % (chains, name, a, b, a / 20 + b, b, a / 20))
        sim.close()
    small.close()
