"""Debug aid: dump the LDS working set of the joint-tree kernel for one env (library built with -DRB_TREE_DEBUG,
ROBOY_SIM_LIB=gym_roboy_amd/csrc/variants/lib_dbg.so) and compare per link with the fp64 prototype (tests/proto/aba_world.py)."""
import ctypes, sys
sys.path.insert(0, "tests"); sys.path.insert(0, "."); sys.path.insert(0, "tests/proto")
import numpy as np
from random_robots import random_tree_robot
from conftest import random_states
from gym_roboy_amd.envs.simulations import HipBatchSimulation
from oracle.physics_np import TendonRobotOracle
from test_tree_tables import TreeDev

seed, env = 101, int(sys.argv[1]) if len(sys.argv) > 1 else 1
robot, desc = random_tree_robot(seed, n_q=17, n_t=20, shape="chain")
n = 37
q, qd, sp = random_states(desc, n, seed)
sim = HipBatchSimulation(robot, n, integrator="euler")
lib = sim._lib
sim.set_state(q, qd)
assert lib.rb_debug_tree_arm(sim.handle, ctypes.c_long(env)) == 0
sim.forward_step_command(sp)
buf = np.zeros(1 << 16, np.float32)
dev = TreeDev()
assert lib.rb_debug_tree_fetch(sim.handle, buf.ctypes.data_as(ctypes.c_void_p), len(buf), ctypes.byref(dev)) == 0
ES, LS = dev.ES, 37
blk = buf[(env % 2) * ES:(env % 2 + 1) * ES]
src = open("tests/proto/aba_world.py").read().split("if __name__")[0].replace("    return qdd", "    return qdd, D, u_, U, a, s, c, fext")
ns = {}
exec(src, ns)
orc = TendonRobotOracle(desc)
qdd, D, u_, U, a, s, c, fext = ns["accel"](desc, orc, q[env].astype(np.float64), qd[env].astype(np.float64), sp[env].astype(np.float64))
np.set_printoptions(precision=4, suppress=True, linewidth=220)
print("ES", ES, "n_levels", dev.n_levels, "o_W", dev.o_W)
for i in range(desc.n_q):
    b = blk[i * LS:(i + 1) * LS]
    kU, kinvD, ku, kqdd, ka, ks, kc, kpT = b[0:6], b[6], b[7], b[8], b[9:15], b[18:24], b[24:30], b[30:36]
    flag = "" if np.all(np.isfinite(b[:36])) else "  <-- non-finite"
    print("link %2d  invD %.5g (ref %.5g)  u %.5g (ref %.5g)  qdd %.5g (ref %.5g)%s" % (i, kinvD, 1 / D[i], ku, u_[i], kqdd, qdd[i], flag))
    print("     U   ", kU, " ref", U[i])
    print("     s   ", ks, " ref", s[i])
    print("     pT  ", kpT, " ref", -fext[i])
