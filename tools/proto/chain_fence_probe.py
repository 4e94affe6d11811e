#!/usr/bin/env python3
"""Probe (round 6): which fence scopes do the packets of a two-chain rollout carry?  Run with AMD_LOG_LEVEL=4 (ROCclr writes every AQL
packet it enqueues to stderr: header acquire / release scopes of kernel dispatches, barrier packets of event records and stream waits).
Markers on stderr delimit the phases so that the log can be cut:
    A  the first 8-step rollout (graphs captured + instantiated + replayed, fork and join)
    B  a second 8-step rollout (graphs replayed only)
    C  a 22-step rollout: eager head (one ring turn of plain launches on both chains), a 16-step graph, two trailing eager steps
    D  read_state (pack kernel behind the join on the handle's stream, copy, synchronise)
usage: AMD_LOG_LEVEL=4 python3 tools/proto/chain_fence_probe.py 2> log"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from gym_roboy_amd.envs.robots import MsjRobot
from gym_roboy_amd.envs.simulations import HipBatchSimulation


def mark(text):
    torch.cuda.synchronize()
    sys.stderr.write("=== PROBE %s ===\n" % text)
    sys.stderr.flush()


n = 262144
st = torch.cuda.Stream()
sim = HipBatchSimulation(MsjRobot(), n, integrator="rk4", seed=1)
sim.set_stream(st.cuda_stream)
assert sim.rollout_chains() == 2
ring = torch.empty(4 * n * 8, dtype=torch.float32, device="cuda")
for r in range(4):
    sim.fill_actions_dev(ring.data_ptr() + 4 * r * n * 8, r)
mark("A begin: first 8-step rollout (capture + replay)")
sim.rollout_dev(ring.data_ptr(), 4, 8, 0.3, use_graph=True)
mark("A end / B begin: second 8-step rollout (replay only)")
sim.rollout_dev(ring.data_ptr(), 4, 8, 0.3, use_graph=True)
mark("B end / C begin: 22-step rollout (eager head, 16-step graphs, 2 trailing steps)")
sim.rollout_dev(ring.data_ptr(), 4, 22, 0.3, use_graph=True)
mark("C end / D begin: read_state")
q, qd, feas = sim.read_state()
mark("D end")
sim.close()
print("probe done; event flags env ROBOY_SIM_EVENT_SYSTEM_FENCE=%r" % os.environ.get("ROBOY_SIM_EVENT_SYSTEM_FENCE"))
