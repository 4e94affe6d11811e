#!/usr/bin/env python3
"""Diagnostic (round 5): where a launch of the one-wave joint-tree kernels (tree_lane.hpp) spends its time at large batches.
Needs a -DRB_LANE_STAMPS build of the library (ROBOY_SIM_LIB=.../variants/lib_lane_stamps.so): every wave records the
constant 100 MHz clock and the shader clock at its entry, behind its input rows, behind the step and behind its output rows.
Prints, for the LAST of a few launches: the spread of the waves' entry times, and per phase the median / 5 % / 95 % duration
in microseconds, plus the launch's extent (first entry -> last exit).

    ROBOY_SIM_LIB=$PWD/gym_roboy_amd/csrc/variants/lib_lane_stamps.so python tools/lane_stamps.py [euler|rk4] [n_envs] [env] [chains]
"""
import ctypes
import sys

import numpy as np

sys.path.insert(0, "/root/repo")
import torch  # noqa: E402
from gym_roboy_amd import _native as nat  # noqa: E402
from gym_roboy_amd.envs.robots import UpperBodyRobot  # noqa: E402
from gym_roboy_amd.envs.simulations import HipBatchSimulation  # noqa: E402

integ = sys.argv[1] if len(sys.argv) > 1 else "euler"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
env_layer = len(sys.argv) > 3 and sys.argv[3] == "env"
act = (torch.rand((n, 38), device="cuda") * 2 - 1).contiguous()
if env_layer:
    from gym_roboy_amd.envs.vec_env import RoboyVecEnv
    env = RoboyVecEnv(UpperBodyRobot(), n, integrator=integ)
    sim = env.sim
    obs = torch.empty((n, 60), device="cuda"); rew = torch.empty(n, device="cuda"); done = torch.empty(n, dtype=torch.int32, device="cuda")
    step = lambda: env.step_dev(act.data_ptr(), obs.data_ptr(), rew.data_ptr(), done.data_ptr())
else:
    sim = HipBatchSimulation(UpperBodyRobot(), n, integrator=integ)
    sim.select_kernel(1)
    step = lambda: sim.step_dev(act.data_ptr(), 0.3)
for _ in range(30):
    step()
sim.synchronize()
waves = min((n + 63) // 64, 4096)
buf = (ctypes.c_ulonglong * (waves * 8))()
lib = nat.load()
lib.rb_debug_lane_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert lib.rb_debug_lane_stamps(buf, waves * 8) == 0
a = np.array(buf, dtype=np.uint64).reshape(waves, 4, 2).astype(np.int64)
real = a[:, :, 0] * 0.01                        # 100 MHz -> us
shader = a[:, :, 1]
t0 = real[:, 0].min()
real -= t0
pct = lambda x: "median %.2f  p5 %.2f  p95 %.2f  max %.2f" % (np.median(x), np.percentile(x, 5), np.percentile(x, 95), x.max())
print("%s %d envs%s: %d waves stamped" % (integ, n, " (fused env layer)" if env_layer else "", waves))
print("  entry after the first wave's  : " + pct(real[:, 0]))
print("  input rows (load + transpose) : " + pct(real[:, 1] - real[:, 0]))
print("  step (acceleration, integrator): " + pct(real[:, 2] - real[:, 1]))
print("  output rows                   : " + pct(real[:, 3] - real[:, 2]))
print("  exit after the first entry    : " + pct(real[:, 3]))
cyc = (shader[:, 2] - shader[:, 1]).astype(np.float64)
us = real[:, 2] - real[:, 1]
print("  shader clock during the step  : %.0f MHz (median), step = %.0f shader cycles (median)" % (np.median(cyc / np.maximum(us, 1e-6)), np.median(cyc)))
# how the phases of different waves overlap: at every 0.25 us, the number of waves in each phase
edges = np.arange(0.0, real[:, 3].max() + 0.25, 0.25)
rows = []
for t in edges:
    rows.append((t, int(((real[:, 0] <= t) & (t < real[:, 1])).sum()), int(((real[:, 1] <= t) & (t < real[:, 2])).sum()),
                 int(((real[:, 2] <= t) & (t < real[:, 3])).sum())))
print("  t [us]: waves loading / stepping / storing")
for t, l, c, s in rows[::4]:
    print("   %5.2f: %5d %5d %5d" % (t, l, c, s))
sim.close()
