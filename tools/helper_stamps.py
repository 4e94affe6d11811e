"""Diagnostic (round 4): barrier stamps of the split joint-tree kernels' workgroup 0 from a -DRB_SPLIT_STAMPS build of the library
(ROBOY_SIM_LIB): per wave, s_memtime before / after every workgroup barrier of ONE step, printed as microseconds since the
first wave's first stamp (s_memtime ticks at 100 MHz on gfx950: 10 ns)."""
import ctypes, sys
import numpy as np
sys.path.insert(0, "/root/repo")
import torch
from gym_roboy_amd import _native as nat
from gym_roboy_amd.envs.robots import UpperBodyRobot
from gym_roboy_amd.envs.simulations import HipBatchSimulation

integ = sys.argv[1] if len(sys.argv) > 1 else "euler"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
env_layer = len(sys.argv) > 3 and sys.argv[3] == "env"
act = (torch.rand((n, 38), device="cuda") * 2 - 1).contiguous()
if env_layer:                                       # the fused env step (tree_split_env_step) instead of the plain step
    from gym_roboy_amd.envs.vec_env import RoboyVecEnv
    env = RoboyVecEnv(UpperBodyRobot(), n, integrator=integ)
    sim = env.sim
    obs = torch.empty((n, 60), device="cuda"); rew = torch.empty(n, device="cuda"); done = torch.empty(n, dtype=torch.int32, device="cuda")
    for _ in range(50):
        env.step_dev(act.data_ptr(), obs.data_ptr(), rew.data_ptr(), done.data_ptr())
else:
    sim = HipBatchSimulation(UpperBodyRobot(), n, integrator=integ)
    for _ in range(50):
        sim.step_dev(act.data_ptr(), 0.3)
sim.synchronize()
buf = (ctypes.c_ulonglong * (8 * 128))()
lib = nat.load()
lib.rb_debug_stamps.argtypes = [ctypes.c_void_p]
assert lib.rb_debug_stamps(buf) == 0
a = np.array(buf, dtype=np.uint64).reshape(8, 128)
t0 = min(int(a[w, 0]) for w in range(8) if a[w, 0])
for w in range(8):
    row = [int(x) for x in a[w] if x]
    if not row:
        continue
    # entries: 0 wave start, 1 rows in LDS, then (before, after) per barrier, then share done, then stores issued
    us = [(x - t0) * 0.01 for x in row[:2 + 2 * 12 + 6]]
    print("wave %d (%d stamps): " % (w, len(row)) + " ".join("%.2f" % u for u in us))
sim.close()
