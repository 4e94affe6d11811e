#!/usr/bin/env python3
"""Probe (round 5): do the consumers behind a two-chain rollout's join see the other chain's writes when the fork / join events carry
no fence of their own (a -DRB_CHAIN_EVENT_FLAGS="(hipEventDisableTiming|hipEventDisableSystemFence)" build, ROBOY_SIM_LIB)?
A batch whose second half starts at a block index that is NOT a multiple of 8, so that the kernel that reads the state afterwards
(pack_state_kernel, another block -> XCD map than the range launch) runs on other XCDs than the writers: stale lines in an
XCD-private L2 would show as a mismatch against the same rollout stepped as one chain.  Prints mismatching iterations."""
import sys
import numpy as np
sys.path.insert(0, "/root/repo")
import torch
from gym_roboy_amd.envs.robots import MsjRobot
from gym_roboy_amd.envs.simulations import HipBatchSimulation

n = 262144 + 768
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 300
robot = MsjRobot()
st = torch.cuda.Stream()
sims = []
for chains in (2, 1):
    s = HipBatchSimulation(robot, n, integrator="rk4", seed=1)
    s.set_stream(st.cuda_stream)
    s.set_rollout_chains(chains)
    sims.append(s)
ring = torch.empty(4 * n * 8, dtype=torch.float32, device="cuda")
for r in range(4):
    sims[0].fill_actions_dev(ring.data_ptr() + 4 * r * n * 8, r)
bad = 0
for it in range(iters):
    outs = []
    for s in sims:
        s.rollout_dev(ring.data_ptr(), 4, 8, 0.3, use_graph=True)
        outs.append(s.read_state())            # pack kernel on the handle's stream right behind the join, then a copy
    same = all(np.array_equal(a, b) for a, b in zip(*outs))
    if not same:
        bad += 1
        d = np.abs(outs[0][0] - outs[1][0]).max()
        print("iteration %d: MISMATCH, max |dq| = %g, first bad env %d" % (it, d, int(np.argmax(np.any(outs[0][0] != outs[1][0], axis=1)))))
        if bad > 5:
            break
print("chains of sim 0: %d; %d iterations, %d mismatches" % (sims[0].rollout_chains(), it + 1, bad))
for s in sims:
    s.close()
sys.exit(1 if bad else 0)
