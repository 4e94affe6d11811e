"""TEST / MEASUREMENT INFRASTRUCTURE - flops per env step from the instrumented
restatement (oracle/flop_count.cpp; SURVEY.md §8(d) "emit the exact count from an
instrumented oracle").

    python -m oracle.flop_count            # prints the table and rewrites profiles/flops_per_env_step.json

bench.py reads the committed JSON (it must not execute anything under oracle/ in
its timed or reporting legs other than cpu_baseline); tests/test_flop_count.py
checks that the JSON is what this module produces and that the instrumented run
reproduces the C oracle's step.
"""
import ctypes
import json
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
JSON_PATH = os.path.join(os.path.dirname(_HERE), "profiles", "flops_per_env_step.json")
FIELDS = ("add", "mul", "div", "minmax", "trans")


def _lib():
    path = os.path.join(_HERE, "libflopcount.so")
    if not os.path.exists(path):
        subprocess.check_call(["make", "-C", _HERE, "libflopcount.so"], stdout=subprocess.DEVNULL)
    return ctypes.CDLL(path)


def count_msj_step(desc, integrator, q, qd, sp, step_size=0.1, n_substeps=1):
    """One instrumented env step of the ball-joint closed form.  Returns
    (counts dict, q', qd', feasible)."""
    return _count(_lib().fc_msj_step, desc, integrator, q, qd, sp, step_size, n_substeps)


def count_tree_step(desc, integrator, q, qd, sp, step_size=0.1, n_substeps=1):
    """One instrumented env step of the joint-tree form (articulated-body algorithm in world
    coordinates, tendons as link crossings: what csrc/tree_aba.hpp evaluates)."""
    return _count(_lib().fc_tree_step, desc, integrator, q, qd, sp, step_size, n_substeps)


def _count(fn, desc, integrator, q, qd, sp, step_size, n_substeps):
    q = np.array(q, dtype=np.float64)
    qd = np.array(qd, dtype=np.float64)
    sp = np.ascontiguousarray(sp, dtype=np.float64)
    out = np.zeros(5, np.uint64)
    feas = ctypes.c_ubyte(0)
    rc = fn(ctypes.byref(desc.as_c_struct()), ctypes.c_double(step_size), int(n_substeps),
                         int(integrator), q.ctypes.data_as(ctypes.c_void_p), qd.ctypes.data_as(ctypes.c_void_p),
                         sp.ctypes.data_as(ctypes.c_void_p), out.ctypes.data_as(ctypes.c_void_p), ctypes.byref(feas))
    if rc:
        raise RuntimeError("flop-count step failed: %d" % rc)
    counts = {k: int(v) for k, v in zip(FIELDS, out)}
    counts["flops"] = sum(counts[k] for k in FIELDS)
    # fewest VALU instructions that can carry these operations: every add fused with a multiply where
    # both exist (v_fma), one instruction per selection and per transcendental
    counts["valu_instr_lower_bound"] = max(counts["add"], counts["mul"]) + counts["div"] + counts["minmax"] + counts["trans"]
    return counts, q, qd, bool(feas.value)


def table():
    from gym_roboy_amd.envs.robots import MsjRobot
    desc = MsjRobot().get_description()
    rng = np.random.default_rng(0)
    q = rng.uniform(0.5 * desc.q_lo, 0.5 * desc.q_hi)
    qd = rng.uniform(-0.5 * desc.qd_max, 0.5 * desc.qd_max)
    sp = rng.uniform(-0.3, 0.3, desc.n_t)
    out = {"_about": "floating-point operations per env step, counted by oracle/flop_count.cpp (MsjRobot: the kernels' closed "
                     "form instantiated with a tallying scalar; UpperBodyRobot: a scalar restatement of tree_aba.hpp's algorithm; "
                     "UpperBodyRobot/*/lane: the operations the generated env-per-lane code executes, constants folded); flops = add + mul + div + minmax + trans; "
                     "regenerate with `python -m oracle.flop_count`"}
    for name, integ in (("euler", 0), ("rk4", 1)):
        c, *_ = count_msj_step(desc, integ, q, qd, sp)
        out["MsjRobot/%s" % name] = c
    from gym_roboy_amd.envs.robots import UpperBodyRobot
    desc = UpperBodyRobot().get_description()
    q = rng.uniform(0.5 * desc.q_lo, 0.5 * desc.q_hi)
    qd = rng.uniform(-0.5 * desc.qd_max, 0.5 * desc.qd_max)
    sp = rng.uniform(-0.3, 0.3, desc.n_t)
    for name, integ in (("euler", 0), ("rk4", 1)):
        c, *_ = count_tree_step(desc, integ, q, qd, sp)
        out["UpperBodyRobot/%s" % name] = c
    # The env-per-lane kernels run code generated for the robot with every constant folded (csrc/tree_lane_gen.hpp):
    # they execute fewer operations than the general algorithm counted above, and their VALU fraction is priced with
    # what they execute.  accel_flops = arithmetic statements of one generated acceleration (the generator's own
    # count); per joint the integrator adds 10 (Euler: v + h a, saturation, q + h v, limit tests) or 26 (RK4: four
    # stage states and saturations, the two weighted sums, the final update, limit tests), per tendon 2 (set-point scaling).
    import sys
    root = os.path.dirname(_HERE)
    sys.path.insert(0, os.path.join(root, "tools"))
    import gen_tree_lane_baked as gen
    build = os.path.join(root, "tests", "_build")
    os.makedirs(build, exist_ok=True)
    accel = gen.generate(desc, os.path.join(build, "flop_count_lane.hpp"))[3]
    for name, evals, per_joint in (("euler", 1, 10), ("rk4", 4, 26)):
        out["UpperBodyRobot/%s/lane" % name] = {"accel_flops": accel, "accel_evaluations": evals,
                                                "flops": evals * accel + per_joint * desc.n_q + 2 * desc.n_t}
    return out


if __name__ == "__main__":
    t = table()
    with open(JSON_PATH, "w") as fh:
        json.dump(t, fh, indent=1, sort_keys=True)
        fh.write("\n")
    print(json.dumps(t, indent=1, sort_keys=True))
