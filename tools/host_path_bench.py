"""PCIe-inclusive rate of the host-array entry point rb_step (plumbing path)."""
import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np
from gym_roboy_amd.envs.robots import MsjRobot
from gym_roboy_amd.envs.simulations import HipBatchSimulation, HipSimulationClient
for n in (4096, 262144):
    sim = HipBatchSimulation(MsjRobot(), n)
    a = np.random.default_rng(0).uniform(-0.3, 0.3, (n, 8)).astype(np.float32)
    for _ in range(3): sim.forward_step_command(a)
    t0 = time.perf_counter(); k = 20
    for _ in range(k): sim.forward_step_command(a)
    dt = (time.perf_counter() - t0) / k
    print("rb_step host path n=%d: %.1f us per call, %.3e env-steps/s (H2D 32 B + D2H 25 B per env)" % (n, dt * 1e6, n / dt))
    sim.close()
c = HipSimulationClient(MsjRobot())
t0 = time.perf_counter(); k = 2000
for _ in range(k): c.forward_step_command([0.1] * 8)
print("HipSimulationClient single env: %.1f us per step" % ((time.perf_counter() - t0) / k * 1e6))
