#!/bin/bash
# Round 2: matrix-core routing (msj_mfma.hpp) against the vector form: agreement, then timings
set -o pipefail
cd /root/repo
mkdir -p gpurun_out
timeout -k 10 300 python - <<'PY' || exit 1
import os, subprocess, sys, json
code = r'''
import numpy as np, sys
from gym_roboy_amd.envs.robots import MsjRobot
from gym_roboy_amd.envs.simulations.hip_simulation_client import HipBatchSimulation
n = 100_003
out = {}
for integ in ("euler", "rk4"):
    sim = HipBatchSimulation(MsjRobot(), n, integrator=integ, seed=1)
    rng = np.random.default_rng(0)
    for t in range(12):
        q, qd, feas = sim.forward_step_command(rng.uniform(-0.3, 0.3, (n, 8)).astype(np.float32))
    out[integ] = (q, qd, feas)
    print(integ, "feasible", feas.mean(), flush=True)
np.savez(sys.argv[1], **{k + "_" + nm: v for k, t in out.items() for nm, v in zip(("q", "qd", "f"), t)})
'''
import numpy as np
for tag, env in (("valu", "0"), ("mfma", "1")):
    e = dict(os.environ, ROBOY_SIM_MFMA=env)
    subprocess.check_call([sys.executable, "-c", code, "/tmp/state_%s.npz" % tag], env=e)
a, b = np.load("/tmp/state_valu.npz"), np.load("/tmp/state_mfma.npz")
for k in a.files:
    d = np.abs(a[k].astype(np.float64) - b[k].astype(np.float64)).max()
    print(k, "max |valu - mfma| =", d, "max |valu| =", np.abs(a[k]).max())
    assert d < 2e-5, k
print("agreement ok")
PY
for M in 0 1; do
 for W in msj-262144-rk4 msj-2097152-euler msj-262144-euler; do
  ROBOY_SIM_MFMA=$M timeout -k 10 200 python bench.py --workload $W --no-also --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('mfma=$M $W', 'launch_us', round(d['roofline']['launch_us_events'],2), 'value', '%.3e'%d['value'])" || exit 1
 done
done
