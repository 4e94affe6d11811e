#!/usr/bin/env python3
"""bench.py - env-steps/s of the batched MsjRobot physics step on N MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload NAME]

One "step" = one lock-step advance of every env of the rank's shard by one
integrator step of dt = 0.1: the batched ``forward_step_command``.
Per step and chain one kernel launch: the default rollout steps large batches as
two independent chains of launches, one per half of the batch (``rb_rollout_chains``;
``roofline.launches_per_step``), and the line carries the figure for ONE launch per
step over the whole batch - what a per-step caller (``rb_step_dev``) gets - beside
it (``roofline.one_launch``).  Actions are i.i.d. U[-1,1) from the Philox
streams, pre-generated as a ring of 4 slabs resident in HBM and rescaled by
0.3 in the kernel.

Default workload for every --gpus N: 262 144 envs per GPU, RK4 (BASELINE.json
configs[2] at N = 1; at N = 8 the same shard size as configs[4], 2 097 152 envs
in all).  The other single-GPU configs (4 096 envs, the Euler shard, a 2 097 152
env batch, the upper body, the fused env layer) are measured briefly at N = 1;
each contributes ONE compact row to ``roofline.configs``
([us per step, HBM fraction, vector-fp32 fraction, launches per step]) and its
long form goes to ``bench_also.json`` beside this file and to stderr.

Output: the LAST thing on stdout is ONE compact JSON line of at most LINE_CAP =
4 096 bytes with the contract keys (metric, value, unit, n_gpus, steps, warmup,
repeats, ms_per_step, higher_is_better, scaling, vs_baseline, dtype, data,
config, roofline, cpu_baseline, collective, sanity); a longer line is a bug and
the exit code says so (4).  Strict JSON: non-finite numbers are written as null.

Timing (``timing_protocol`` 3): W untimed warm-up steps, then the K-step region -
bracketed by a barrier and a device synchronisation on both sides - is timed 2 R
times (R chosen so that R repeats cover >= 50 ms of device time, so a small K
does not turn the number into a measurement of launch latency).  The regions
alternate: R are timed by the wall clock alone, R carry two HIP events on the
launch stream around their K launches as well.  ``ms_per_step`` is the MEAN over
the event-free repeats of (max over ranks of the region's wall time) / K
(``ms_per_step_median`` beside it, ``ms_per_step_with_events`` = the same mean
over the other R) and ``value`` the env steps of all ranks in one region divided
by that time; ``roofline`` is built from the events (median).  (Protocol 2,
rounds 4-5, recorded the events in every region: the two records - markers with
a system-scope fence - cost a 20-step region 5-10 us of its wall time, which is
instrumentation, not the workload.  Protocol 1, rounds 1-3, printed the median
and reduced the statistics in every region.)

N > 1: one rank per GPU under torch.distributed.run - started by the caller (WORLD_SIZE set: behave as a rank) or, when
`python bench.py --gpus N` is run bare, by bench.py itself as a fresh child process tree before any GPU call
(``self_launch``: rank 0's line is relayed, the exit code is the launcher's); envs shard
contiguously (weak scaling: the per-GPU batch is fixed), no data-path
collective; every STATS_EVERY = 100 steps (counted across the timed regions) the
rank's episode statistics (8 doubles) are reduced on the device and all-reduced
over RCCL, and the line carries a ``collective`` object that audits it.  The
one-launch-per-step leg (``roofline.one_launch``) runs at N = 1, at N > 1 only
with --one-launch.  Rank 0 prints the line.
"""
import argparse
import json
import math
import os
import statistics
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

WORKLOADS = {
    # name: (envs per GPU, integrator, substeps, default steps, default warmup, BASELINE.json config)
    "msj-262144-rk4": (262144, "rk4", 1, 400, 40, "configs[2]: MsjRobot 262 144 envs per GPU, RK4 fp32 (x8 GPUs = configs[4]'s 2 097 152 envs)"),
    "msj-262144-euler": (262144, "euler", 1, 800, 80, "configs[4] shard: MsjRobot 262 144 envs per GPU, semi-implicit Euler fp32"),
    "msj-4096-euler": (4096, "euler", 1, 4000, 200, "configs[1]: MsjRobot 4 096 envs, semi-implicit Euler fp32"),
    "msj-2097152-euler": (2097152, "euler", 1, 300, 30, "large batch: MsjRobot 2 097 152 envs on one GPU, Euler fp32"),
    "msj-2097152-rk4": (2097152, "rk4", 1, 100, 10, "large batch: MsjRobot 2 097 152 envs on one GPU, RK4 fp32 (the headline kernel at 8x its waves per SIMD)"),
    "upper-body-8192-euler": (8192, "euler", 1, 300, 30, "configs[3]: upper body (20 DOF / 38 tendons) 8 192 envs, Euler fp32"),
    "upper-body-8192-rk4": (8192, "rk4", 1, 100, 10, "configs[3]: upper body (20 DOF / 38 tendons) 8 192 envs, RK4 fp32"),
    "upper-body-65536-euler": (65536, "euler", 1, 300, 30, "large batch: upper body (20 DOF / 38 tendons) 65 536 envs (one wave per SIMD), Euler fp32"),
}
DEFAULT_WORKLOAD = "msj-262144-rk4"
RING = 4
# Steps between two all-reduces of the statistics block: K = 100, the measurement spec of SURVEY.md §8(d) config (5)
# (a quarter of an episode horizon: RoboyEnv.max_episode_length = 400 env steps, roboy_env.py:23).
STATS_EVERY = int(os.environ.get("ROBOY_BENCH_STATS_EVERY", "100"))
MIN_TIMED_S = float(os.environ.get("ROBOY_BENCH_MIN_TIMED_S", "0.05"))   # device time the repeats must cover
MAX_REPEATS = 256
HBM_PEAK = 8.0e12          # B/s, spec (MI355X_MICROARCH.md "HBM3E peak BW")
HBM_COPY = 6.29e12         # B/s, measured float4 copy (same table)
VALU_PEAK = 157.3e12       # flop/s, fp32 vector peak (MI355X_MICROARCH.md "Peak FP32 (vector)")
# measured rate of independent v_fma_f32 on this chip: 1.20 ns per wave-instruction and SIMD at 8 waves per SIMD
# (profiles/r1_b/microbench_valu.log; v_pk_fma_f32 takes twice as long, i.e. the same flop rate): 64 lanes x 2 flop
# x 1024 SIMDs / 1.20 ns.  The spec peak is the roof `frac` is quoted against; this is what an all-FMA stream reaches.
VALU_FMA_MEASURED = 64 * 2 * 1024 / 1.20e-9
KERNEL_NAMES = {1: "msj_step_env_per_lane", 2: "msj_step_tendon_per_lane", 3: "tree_step_aba", 5: "msj_step_mirror_pairs"}
TREE_KERNEL_NAMES = {1: "tree_lane_step", 3: "tree_step_aba", 4: "tree_split_step", 6: "tree_split_step"}   # joint trees: env-per-lane (generated) / octets / several waves per env group
PROFILE_DIRS = ("r6_a", "r5_a", "r4_a", "r3_a", "r2_a", "r1_b")   # newest first: where the committed rocprofv3 PMC passes live
GENERATED_HEADERS = ("tree_lane_baked.hpp", "tree_lane_split_baked.hpp", "tree_lane_split2_baked.hpp")   # `make` output, git-ignored


def csrc_hash():
    """16 hex digits over everything the kernels are built from: the tracked sources under gym_roboy_amd/csrc (the generated
    headers are functions of them), the robot descriptions and the two generator scripts.  Profile summaries are stamped with it
    at collection (tools/summarize_profile.py); a stamp that differs from the tree's hash marks a counter row as stale."""
    import glob
    import hashlib
    files = []
    for pat in ("gym_roboy_amd/csrc/*.hip", "gym_roboy_amd/csrc/*.hpp", "gym_roboy_amd/csrc/*.cpp", "gym_roboy_amd/csrc/Makefile",
                "gym_roboy_amd/envs/robots/data/*.json", "tools/gen_tree_lane_baked.py", "tools/gen_msj_baked.py"):
        files += glob.glob(os.path.join(ROOT, pat))
    h = hashlib.sha256()
    for f in sorted(set(files)):
        if os.path.basename(f) in GENERATED_HEADERS:
            continue
        h.update(os.path.relpath(f, ROOT).encode() + b"\0")
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


_PMC_CACHE = {}


def pmc_traffic(workload):
    """(HBM bytes per launch, source file, stale) of the step kernel from the committed rocprofv3 PMC passes
    (profiles/<round>/hbm_traffic_pmc.json: separate --pmc FETCH_SIZE and --pmc WRITE_SIZE runs of this workload, FETCH_SIZE
    doubled as the gfx950 guide prescribes).  PMC collection cannot run inside the timed bench, so the number is the profiled
    one, not a live one.  The NEWEST profile directory that holds the workload answers; `stale` is True unless that entry was
    stamped, at collection, with the hash of the kernel sources this tree has (csrc_hash) - an older round's row is never passed
    off as this library's.  (None, None, None) if the workload was never profiled."""
    if "hash" not in _PMC_CACHE:
        _PMC_CACHE["hash"] = csrc_hash()
    for d in PROFILE_DIRS:
        try:
            with open(os.path.join(ROOT, "profiles", d, "hbm_traffic_pmc.json")) as fh:
                entry = json.load(fh)[workload]
            return entry["hbm_bytes_per_launch"], "profiles/%s/hbm_traffic_pmc.json" % d, entry.get("csrc_hash") != _PMC_CACHE["hash"]
        except Exception:
            continue
    return None, None, None


def flops_per_env_step(robot_name, integrator, substeps, kernel=None):
    """Floating-point operations of one env step, from the committed count of the
    instrumented restatement (oracle/flop_count.cpp -> profiles/flops_per_env_step.json);
    None for a robot that has not been counted.  The generated env-per-lane kernels of the
    joint-tree robots execute fewer operations than the general algorithm (constants folded):
    they are priced with their own count ("<robot>/<integrator>/lane")."""
    try:
        with open(os.path.join(ROOT, "profiles", "flops_per_env_step.json")) as fh:
            table = json.load(fh)
        key = "%s/%s" % (robot_name, integrator)
        if kernel in ("tree_lane_step", "tree_split_step") and key + "/lane" in table:
            key += "/lane"
        return table[key]["flops"] * substeps
    except Exception:
        return None


def roofline(robot_name, integrator, substeps, n_envs, bytes_per_env_step, launch_s, workload=None, kernel=None, chains=1):
    """Both candidate roofs for one STEP of the step kernel over the batch (``launch_s`` = timed region / steps); ``bound``
    names the one the kernel sits closer to (the larger fraction).  chains = 2: a step is two concurrent launches over half
    the batch each (rb_rollout_dev's graphs, large ball-joint batches); rocprofv3 then lists half-batch launches whose own
    durations overlap - the per-step time, not a kernel's duration, is what the fractions are computed from."""
    nbytes = bytes_per_env_step * n_envs
    gbps = nbytes / launch_s / 1e9
    hbm = {"achieved": gbps, "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": gbps * 1e9 / HBM_PEAK,
           "frac_of_measured_copy_6.29TBps": gbps * 1e9 / HBM_COPY,
           "bytes_per_env_step": bytes_per_env_step, "bytes_per_launch": nbytes}
    flops = flops_per_env_step(robot_name, integrator, substeps, kernel)
    valu = None
    if flops is not None:
        tf = flops * n_envs / launch_s / 1e12
        valu = {"achieved": tf, "peak": VALU_PEAK / 1e12, "unit": "TFLOP/s", "frac": tf * 1e12 / VALU_PEAK,
                "frac_of_measured_fma_rate_109TFLOPs": tf * 1e12 / VALU_FMA_MEASURED,
                "flops_per_env_step": flops, "flops_per_launch": flops * n_envs,
                "source": "profiles/flops_per_env_step.json (instrumented restatement, oracle/flop_count.cpp)"}
    top = valu if (valu is not None and valu["frac"] > hbm["frac"]) else hbm
    traffic, src, stale = pmc_traffic(workload) if workload else (None, None, None)
    return {"bound": "valu" if top is valu else "hbm", "achieved": top["achieved"], "peak": top["peak"],
            "unit": top["unit"], "frac": top["frac"], "traffic": traffic, "traffic_source": src, "traffic_stale": stale,
            "launch_us_events": launch_s * 1e6, "launches_per_step": chains, "envs_per_launch": n_envs // chains if chains > 1 else n_envs,
            "hbm": hbm, "valu": valu}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--workload", default=DEFAULT_WORKLOAD, choices=sorted(WORKLOADS))
    ap.add_argument("--envs", type=int, default=None, help="override envs per GPU")
    ap.add_argument("--substeps", type=int, default=None, help="override integrator substeps per env step")
    ap.add_argument("--kernel", type=int, default=0, help="0 auto, 1 env-per-lane, 2 tendon-per-lane (ball joints), 3 octets (joint trees), 4 env-per-lane split over several waves (joint trees), 5 two lanes per env (ball joints with a mirror plane), 6 the lean two-part split (joint trees, two workgroups per CU)")
    ap.add_argument("--no-graph", action="store_true", help="eager per-step launches instead of hipGraph replay")
    ap.add_argument("--no-also", action="store_true", help="skip the secondary workloads")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--one-launch", action="store_true", help="N > 1: also time the headline with one launch per step over the whole shard (always on at N = 1)")
    ap.add_argument("--cpu-seconds", type=float, default=16.0)
    ap.add_argument("--repeats", type=int, default=None, help="fix the number of timed repeats (default: cover 50 ms)")
    return ap.parse_args()


def run_workload(torch, robot, name, steps, warmup, envs, use_graph, rank, world, dist, substeps=None, kernel=0,
                 repeats=None, force_chains=0):
    """Returns dict(ms_per_step, value, kernel_us, ...) for this workload."""
    from gym_roboy_amd.envs.simulations import HipBatchSimulation
    n_envs, integrator, nsub, d_steps, d_warm, label = WORKLOADS[name]
    n_envs = envs or n_envs
    nsub = substeps or nsub
    steps = steps or d_steps
    warmup = d_warm if warmup is None else warmup
    dev = torch.cuda.current_device()
    sim = HipBatchSimulation(robot, n_envs, integrator=integrator, n_substeps=nsub, device=dev,
                             seed=0, env_id_offset=rank * n_envs)
    sim.select_kernel(kernel)
    if force_chains:
        sim.set_rollout_chains(force_chains)
    stream = torch.cuda.current_stream()        # main() made a non-default stream current
    sim.set_stream(stream.cuda_stream)          # launches and torch events share one stream
    slab = n_envs * sim.n_t
    ring = torch.empty(RING * slab, dtype=torch.float32, device="cuda")
    for r in range(RING):
        sim.fill_actions_dev(ring.data_ptr() + 4 * r * slab, r)
    # Episode statistics: every rank reduces its envs' counters on the device (rb_env_stats_dev) after every
    # STATS_EVERY steps, counted across the timed regions - at N = 1 too, so that every N runs the same
    # per-GPU work and the scaling curve measures the collective alone - and N > 1 all-reduces the block.
    # The all-reduce: 64 bytes, enqueued IN-LINE on the launch stream (a non-async
    # c10d op runs on the current stream).  Measured on one MI355X (profiles/r1_b/
    # rccl_one_rank_rehearsal.log): letting the collective run concurrently on a second
    # stream (async_op=True, or an own side stream) slows the graph-replayed step kernels
    # for the whole rollout, independent of how often it runs; in line it costs its own
    # ~20 us per call and nothing else.
    stats_ring = [torch.zeros(8, dtype=torch.float64, device="cuda") for _ in range(2)]
    state = {"chunk": 0, "last": stats_ring[0], "steps_issued": 0, "calls": 0, "ar_events": [], "since": 0}
    act_scale = float(robot.get_action_space().high[0])

    def reduce_stats():
        from gym_roboy_amd import _native as nat
        import ctypes
        buf = stats_ring[state["chunk"] % 2]
        state["chunk"] += 1
        nat.check(sim._lib.rb_env_stats_dev(sim.handle, ctypes.c_void_p(buf.data_ptr()), 0))
        state["last"] = buf
        state["calls"] += 1
        if dist is None:                                            # one GPU: the local reduction alone
            return
        if dist.get_backend() == "nccl":
            # events around the in-line collective (same stream): lets a scaling line be decomposed
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            dist.all_reduce(buf)                                    # RCCL over xGMI
            e1.record(stream)
            if len(state["ar_events"]) < 64:
                state["ar_events"].append((e0, e1))
        else:                                                       # rehearsal over gloo: through the host
            host = buf.cpu()
            dist.all_reduce(host)
            buf.copy_(host)

    def advance(k, before_reduce=None):
        """k steps; the statistics block is reduced (and all-reduced) every STATS_EVERY steps, counted ACROSS calls - the
        measurement spec's cadence whatever K a timed region has.  before_reduce: called once all k
        steps are issued, in front of a reduction that falls on the region's end (the closing event goes there)."""
        done = 0
        while done < k:
            chunk = min(STATS_EVERY - state["since"], k - done)
            sim.rollout_dev(ring.data_ptr(), RING, chunk, act_scale, use_graph=use_graph)
            done += chunk
            state["steps_issued"] += chunk
            state["since"] += chunk
            if done == k and before_reduce is not None:
                before_reduce()
            if state["since"] >= STATS_EVERY:
                reduce_stats()
                state["since"] = 0

    def timed_region(with_events=True):
        """One K-step region between barrier + synchronize pairs.  Returns (wall seconds = max over ranks, device seconds between
        two events on the launch stream that bracket the K launches and nothing else - None for a region timed by the wall clock
        alone).  The two event records are instrumentation with a price of their own - markers with a system-scope fence in front
        of the first launch and behind the last: the empty bracket (record, record, synchronize) takes 17 us of wall time on this
        stack, profiles/r6_a/rollout_call_overhead.log - so the regions alternate: the wall clock of the event-free ones makes
        `ms_per_step` / `value`, the events of the others make `roofline`."""
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        ev0 = ev1 = None
        if with_events:
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        if with_events:
            ev0.record(stream)
        advance(steps, before_reduce=(lambda: ev1.record(stream)) if with_events else None)
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0          # this rank: start of the region -> its work is complete
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        if dist is not None:
            t = torch.tensor([wall], dtype=torch.float64, device="cuda" if dist.get_backend() == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            wall = float(t.item())
        return wall, (ev0.elapsed_time(ev1) * 1e-3 if with_events else None)

    # 16 untimed steps from the reset state decorrelate the envs, then warm-up
    advance(16)
    advance(warmup)
    # untimed rehearsal of the timed region, so that no hipGraph is captured / instantiated
    # inside it (graphs are cached per chunk size) and RCCL's first call is behind us
    timed_region()
    if repeats is None:
        # one more untimed region sizes the repeats (the rehearsal above paid for graph capture): cover
        # >= MIN_TIMED_S of device time; same R on every rank (the region's wall time is the max over ranks)
        est_wall, _ = timed_region()
        repeats = max(3, min(MAX_REPEATS, int(np.ceil(1.05 * MIN_TIMED_S / max(est_wall, 1e-6)))))
    # 2 R regions, alternating: R timed by the wall clock alone (-> ms_per_step, value), R with the two events as well (-> roofline)
    walls, walls_ev, devs = [], [], []
    for i in range(2 * repeats):
        w, d = timed_region(with_events=(i % 2 == 1))
        if d is None:
            walls.append(w)
        else:
            walls_ev.append(w)
            devs.append(d)
    # The statistics reduction (+ all-reduce at N > 1) falls into one region in every STATS_EVERY / K: the MEAN over the repeats
    # carries its amortised cost (a median would hide it whenever fewer than half of the regions contain one).
    wall = statistics.fmean(walls)
    launch_s = statistics.median(devs) / steps   # HIP events on the launch stream around the K steps (no reduction inside unless K > STATS_EVERY)
    reduce_stats()                               # untimed: the audit below wants a block that covers every step issued
    q, qd, feas = sim.read_state()
    info = sim.info()
    # audit of the collective: the all-reduced block's slot 6 is the sum over ranks of the
    # env steps each rank has issued since its simulation was created
    collective = None
    if dist is not None:
        last = [float(x) for x in state["last"].cpu()]
        expected = float(world) * n_envs * state["steps_issued"]
        collective = {"backend": "rccl (torch.distributed 'nccl')" if dist.get_backend() == "nccl" else dist.get_backend(),
                      "world_size": dist.get_world_size(), "allreduce_calls": state["calls"],
                      "payload_bytes": 64, "every_steps": STATS_EVERY,
                      "us_per_allreduce": (statistics.median([a.elapsed_time(b) * 1e3 for a, b in state["ar_events"][1:]])
                                           if len(state["ar_events"]) > 1 else None),
                      "n_env_steps_allreduced": last[6], "expected": expected,
                      "ok": last[6] == expected and dist.get_world_size() == world}
    stats = [float(x) for x in state["last"].cpu()]
    chains = sim.rollout_chains() if use_graph else 1
    sim.close()
    robot_name = type(robot).__name__
    kernel_name = (TREE_KERNEL_NAMES if robot_name == "UpperBodyRobot" else KERNEL_NAMES)[info["kernel"]]
    return {
        "workload": name, "label": label, "envs_per_gpu": n_envs, "integrator": integrator,
        "substeps": nsub, "steps": steps, "warmup": warmup, "repeats": repeats,
        "value": world * n_envs * steps / wall, "ms_per_step": wall * 1e3 / steps,
        "ms_per_step_min": min(walls) * 1e3 / steps, "ms_per_step_max": max(walls) * 1e3 / steps,
        "ms_per_step_median": statistics.median(walls) * 1e3 / steps,
        "ms_per_step_with_events": statistics.fmean(walls_ev) * 1e3 / steps, "event_repeats": len(devs),
        "timed_device_ms": sum(devs) * 1e3,
        "launch_us_events": launch_s * 1e6,
        "roofline": roofline(robot_name, integrator, nsub, n_envs, info["bytes_per_env_step"], launch_s,
                             name if envs is None and substeps is None else None, kernel_name, chains),
        "kernel": kernel_name,
        "stats": stats, "collective": collective,
        "finite": bool(np.isfinite(q).all() and np.isfinite(qd).all()),
        "feasible_frac": float(feas.mean()),
    }


def run_fused_env(torch, robot, n_envs, steps=300, warmup=30, spread=False, chains=1):
    """Secondary line: the fused env layer (rb_env_step_dev: rescale + physics +
    obs/reward/done/goal, 84 + 72 algorithmic bytes per env step), eager launches.
    spread: the envs' episode counters start uniformly spread over an episode (as in training once goals have been reached
    here and there), so every step ends one episode in 400 - and a launch waits for the waves that handle one; default: all
    episodes in lock-step, none ends inside the timed steps.
    chains = 2: the batch as two halves on two streams (rb_env_step_range_dev), forked once before the timed steps and joined once
    behind them - how a closed-loop caller runs its rollout (gym_roboy_amd/ppo.py): one half's row traffic lies under the other
    half's arithmetic.  Same results (envs are independent)."""
    from gym_roboy_amd.envs.vec_env import RoboyVecEnv
    env = RoboyVecEnv(robot, n_envs)
    if spread:
        rng = np.random.default_rng(3)
        box = robot.get_joint_angles_space()
        env.reset()
        env.set_goal(rng.uniform(box.low, box.high, (n_envs, env.n_q)).astype(np.float32), rng.integers(1, 401, n_envs).astype(np.uint32))
    st = torch.cuda.current_stream()
    env.sim.set_stream(st.cuda_stream)
    acts = [torch.rand((n_envs, env.n_t), device="cuda") * 2 - 1 for _ in range(RING)]
    obs = torch.empty((n_envs, 3 * env.n_q), device="cuda")
    rew = torch.empty(n_envs, device="cuda")
    done = torch.empty(n_envs, dtype=torch.int32, device="cuda")
    if chains == 2 and not env.range_capable():
        chains = 1
    if chains == 2:
        # each chain's steps as replays of a captured graph of GSTEPS range launches (two eager launches per step from Python
        # take longer than the kernels): what gym_roboy_amd/ppo.py does with its (policy step, env step) pairs
        half = ((n_envs // 2 + 255) // 256) * 256
        st2 = torch.cuda.Stream()
        GSTEPS = 12 * RING
        steps = max(GSTEPS, (steps // GSTEPS) * GSTEPS)
        warmup = GSTEPS
        ptrs = (obs.data_ptr(), rew.data_ptr(), done.data_ptr())
        env.step_range_dev(0, half, st.cuda_stream, acts[0].data_ptr(), *ptrs)       # (first call outside any capture)
        torch.cuda.synchronize()
        graphs = []
        for first, count, stream in ((0, half, st), (half, n_envs - half, st2)):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=stream):
                for t in range(GSTEPS):
                    env.step_range_dev(first, count, stream.cuda_stream, acts[t % RING].data_ptr(), *ptrs)
            graphs.append((g, stream))

        def advance(k):
            st2.wait_stream(st)                                  # fork once ...
            for _ in range(k // GSTEPS):
                with torch.cuda.stream(st2):
                    graphs[1][0].replay()
                with torch.cuda.stream(st):
                    graphs[0][0].replay()
            st.wait_stream(st2)                                  # ... join once
    else:
        def advance(k):
            for t in range(k):
                env.step_dev(acts[t % RING].data_ptr(), obs.data_ptr(), rew.data_ptr(), done.data_ptr())
    advance(warmup)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record(st)
    advance(steps)
    e1.record(st)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    us = e0.elapsed_time(e1) * 1e3 / steps
    bytes_per = 4 * (4 * env.n_q + env.n_t + 1) + 4 * (3 * env.n_q + env.n_q + 6)   # + obs, goal, counter x2, return x2, reward, done
    finite = bool(torch.isfinite(obs).all().item() and torch.isfinite(rew).all().item())
    q, qd, feas = env.sim.read_state()
    env.sim_kernel = env.sim.info()["kernel"]
    env.close()
    name = "fused-env-%d" % n_envs if type(robot).__name__ == "MsjRobot" else "fused-env-%s-%d" % (type(robot).__name__, n_envs)
    if spread:
        name += "-spread"
    if chains == 2:
        name += "-chains"
    return {"workload": name, "label": "fused env layer (RoboyVecEnv.step), %s, %d envs, Euler fp32%s" % (type(robot).__name__, n_envs, ", episodes spread out" if spread else ""),
            "value": n_envs * steps / wall, "ms_per_step": wall * 1e3 / steps, "launch_us_events": us, "steps": steps,
            "roofline": roofline(type(robot).__name__, "euler", 1, n_envs, bytes_per, us * 1e-6, name,
                                 "tree_lane_step" if env.sim_kernel == 1 and type(robot).__name__ != "MsjRobot" else None, chains),
            "finite": finite and bool(np.isfinite(q).all() and np.isfinite(qd).all()), "feasible_frac": float(feas.mean())}


def run_fused_rollout(torch, robot, n_envs, steps_per_launch=100, launches=20):
    """Secondary line, never the headline: open-loop rollout with the steps fused
    into one launch (rb_rollout_fused_dev).  State stays in registers, only the
    32-byte action record is read per env step; not the per-step contract
    (no policy can run between steps)."""
    from gym_roboy_amd.envs.simulations import HipBatchSimulation
    sim = HipBatchSimulation(robot, n_envs)
    st = torch.cuda.current_stream()
    sim.set_stream(st.cuda_stream)
    ring = torch.rand((RING, n_envs, sim.n_t), device="cuda") * 2 - 1
    sim.rollout_fused_dev(ring.data_ptr(), RING, steps_per_launch, 0.3)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record(st)
    for _ in range(launches):
        sim.rollout_fused_dev(ring.data_ptr(), RING, steps_per_launch, 0.3)
    e1.record(st)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    steps = steps_per_launch * launches
    us = e0.elapsed_time(e1) * 1e3 / steps
    q, qd, feas = sim.read_state()
    sim.close()
    return {"workload": "fused-rollout-%d" % n_envs,
            "label": "open-loop rollout, %d steps fused per launch, %d envs, Euler fp32 (not the per-step contract)"
                     % (steps_per_launch, n_envs),
            "value": n_envs * steps / wall, "ms_per_step": wall * 1e3 / steps, "us_per_step_events": us,
            "steps": steps, "bytes_per_env_step": 4 * sim.n_t,
            "finite": bool(np.isfinite(q).all() and np.isfinite(qd).all()), "feasible_frac": float(feas.mean())}


def run_ppo_iteration(torch, n_envs, fused, iters=2):
    """The consumer of the path: PPO iterations (128-step rollout through the fused env kernel +
    4 epochs x 4 minibatches) on one GPU, timesteps/s end to end; fused: policy step, GAE and minibatch gradient on the
    matrix cores (include/roboy_policy.h) instead of torch's kernels.  Not the headline metric."""
    from gym_roboy_amd.envs.robots import MsjRobot
    from gym_roboy_amd.envs.vec_env import RoboyVecEnv
    from gym_roboy_amd.ppo import PPO
    env = RoboyVecEnv(MsjRobot(), n_envs)
    agent = PPO(env, ent_coef=0.1, device="cuda", reward_scale=0.01, use_graphs=True, fused_policy=fused, fused_update=fused)
    stats = agent.update(agent.collect())
    torch.cuda.synchronize()
    tc = tu = 0.0
    for _ in range(iters):
        t0 = time.perf_counter(); roll = agent.collect(); torch.cuda.synchronize(); t1 = time.perf_counter()
        stats = agent.update(roll); torch.cuda.synchronize(); t2 = time.perf_counter()
        tc += t1 - t0; tu += t2 - t1
    feasible = float(env.sim.read_state()[2].mean())
    env.close()
    return {"workload": "ppo-%d-%s" % (n_envs, "fused" if fused else "torch"),
            "label": "PPO iteration (128-step rollout + 16 minibatch steps), %d envs, %s" % (
                n_envs, "policy step / GAE / gradient as MFMA kernels" if fused else "torch policy, autograd"),
            "value": iters * agent.n_steps * n_envs / (tc + tu), "unit": "timesteps/s", "rollout_ms": tc / iters * 1e3,
            "update_ms": tu / iters * 1e3, "steps": iters * agent.n_steps,
            "finite": bool(all(math.isfinite(v) for v in stats.values())), "feasible_frac": feasible}


_PY_ENV_WORKER = r"""
import contextlib, io, sys, time
sys.path.insert(0, %(root)r)
import numpy as np
from gym_roboy_amd.envs import RoboyEnv
from gym_roboy_amd.envs.robots import MsjRobot
from oracle.cpu_simulation_client import CpuSimulationClient
from oracle import philox_np as ph
rank = int(sys.argv[1]); seconds = float(sys.argv[2])
robot = MsjRobot()
acts = [a for a in ph.actions(0, np.arange(256, dtype=np.uint64) + 256 * rank, 0, 8)]
with contextlib.redirect_stdout(io.StringIO()):
    env = RoboyEnv(simulation_client=CpuSimulationClient(robot), seed=rank)
    env.reset()
    t0 = time.perf_counter(); k = 0
    while time.perf_counter() - t0 < seconds:
        if env.step(acts[k %% 256])[2]:
            env.reset()
        k += 1
    dt = time.perf_counter() - t0
sys.stdout.write("%%d %%.6f\n" %% (k, dt))
"""


def cpu_python_env_loop(num_cpu, seconds):
    """The reference's own architecture (train_parallel.py:19-29): num_cpu OS processes,
    one Python RoboyEnv each, stepping its own simulator - here the oracle-backed
    CpuSimulationClient in place of the absent ROS/CARDSflow instance.  Children are
    plain `python -c` processes started before this process has touched the GPU."""
    code = _PY_ENV_WORKER % {"root": ROOT}
    env = dict(os.environ, OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1", MKL_NUM_THREADS="1")
    procs = [subprocess.Popen([sys.executable, "-c", code, str(r + 1), str(seconds)], stdout=subprocess.PIPE,
                              stderr=subprocess.DEVNULL, text=True, env=env) for r in range(num_cpu)]
    total, per = 0.0, []
    for p in procs:
        out, _ = p.communicate(timeout=seconds * 10 + 120)
        if p.returncode == 0 and out.strip():
            k, dt = out.split()
            per.append(int(k) / float(dt))
            total += int(k) / float(dt)
    return {"value": total, "processes": len(per), "per_process": (total / len(per)) if per else None,
            "seconds_each": seconds}


def cpu_baseline(robot, seconds, name):
    """The C restatement of the same step (oracle/roboy_oracle.c, fp32 build) timed on this
    box's host cores on a bounded sample of the workload: one thread (the
    scalar port), 64 threads and every core the process may run on (OpenMP over envs; the
    sample is large enough - >= 256 envs per thread - for the threads to pay off); the best of
    them is `value`.  Then the reference's architecture: one Python RoboyEnv per process x num_cpu
    processes.  Runs before the GPU is touched."""
    from oracle.c_oracle import COracle
    from oracle import philox_np as ph
    n_envs, integrator, nsub, *_ = WORKLOADS[name]
    desc = robot.get_description()
    orc = COracle(desc, "f32")
    integ = 0 if integrator == "euler" else 1
    small = desc.n_q <= 3
    cores_avail = len(os.sched_getaffinity(0))
    thread_counts = sorted({1, min(cores_avail, 64), cores_avail})
    budget = seconds * 0.6 / len(thread_counts)
    out = {}
    for threads in thread_counts:
        n = min(n_envs, 4096 if small else 512) if threads == 1 else min(max(n_envs, 256 * threads), 262144 if small else 16384)
        ids = np.arange(n, dtype=np.uint64)
        slabs = [np.ascontiguousarray(ph.actions(0, ids, r, desc.n_t) * np.float32(0.3)) for r in range(RING)]
        q = np.zeros((n, desc.n_q), np.float32)
        qd = np.zeros((n, desc.n_q), np.float32)
        feas = np.zeros(n, np.uint8)
        for t in range(4):
            orc.step_inplace(q, qd, slabs[t % RING], feas, integrator=integ, n_substeps=nsub, threads=threads)
        t0 = time.perf_counter()
        k = 0
        while time.perf_counter() - t0 < budget:
            orc.step_inplace(q, qd, slabs[k % RING], feas, integrator=integ, n_substeps=nsub, threads=threads)
            k += 1
        out[threads] = (n * k / (time.perf_counter() - t0), k, n)
    best = max(out, key=lambda th: out[th][0])
    py_loop = None
    if small:
        num_cpu = min(cores_avail, 64)          # num_cpu = the host cores, capped where the process start-up would eat the budget
        py_loop = cpu_python_env_loop(num_cpu, max(2.0, seconds * 0.25))
        py_loop["cores"] = num_cpu
        py_loop["what"] = ("RoboyEnv.step (Python, one env per process x %d processes, Euler) over the oracle-backed "
                           "CpuSimulationClient: the reference's architecture, train_parallel.py:19-29" % num_cpu)
    return {
        "value": out[best][0], "unit": "env-steps/s", "cores": best, "kind": "port",
        "sample": "%d envs x %d steps of %s, C fp32 restatement (oracle/roboy_oracle.c), %s"
                  % (out[best][2], out[best][1], name, "OpenMP over envs" if best > 1 else "one thread"),
        "by_threads": {str(th): {"value": out[th][0], "envs": out[th][2], "steps": out[th][1]} for th in thread_counts},
        "value_1_core": out[1][0], "host_cores_available": cores_avail,
        "python_env_processes": py_loop,
    }


def brief(r):
    """An `also` entry: the workload's own numbers with both roofline fractions and the sanity fields."""
    keep = ("workload", "label", "value", "ms_per_step", "launch_us_events", "steps", "repeats", "kernel",
            "roofline", "finite", "feasible_frac", "us_per_step_events", "bytes_per_env_step")
    return {k: r[k] for k in keep if k in r}


LINE_CAP = 4096            # bytes: the contract line on stdout never exceeds this (the driver keeps a bounded tail of stdout)
TIMING_PROTOCOL = 3        # 1 (rounds 1-3): statistics reduction at the end of every region, ms_per_step = median of the region walls
                           # 2 (rounds 4-5): reduction every STATS_EVERY steps counted across regions, ms_per_step = mean; the median rides along
                           # 3 (round 6): as 2, and the regions alternate - wall clock alone (-> ms_per_step, value) / wall clock + the two
                           #   HIP events (-> roofline; their wall time rides along as ms_per_step_with_events)
# the secondary workloads whose compact rows ride in roofline.configs (everything else: bench_also.json / stderr)
CONFIG_ROWS = ("msj-4096-euler", "upper-body-8192-euler", "upper-body-8192-rk4", "msj-262144-euler", "msj-2097152-euler",
               "fused-env-2097152", "upper-body-65536-euler", "fused-env-UpperBodyRobot-65536", "fused-env-UpperBodyRobot-65536-chains")
TRAFFIC_ROWS = ("msj-262144-euler", "msj-2097152-euler")    # rows that also carry PMC traffic / algorithmic bytes


def sig(x, digits=5):
    """Floats to `digits` significant digits (the line is capped; 1e-5 relative is far below run-to-run spread); non-finite -> None."""
    if isinstance(x, bool) or x is None or isinstance(x, (int, str)):
        return x
    if isinstance(x, float):
        if not math.isfinite(x):
            return None
        if x == 0.0 or x == int(x) and abs(x) < 1e15:
            return int(x) if x == int(x) else x
        return float("%.*g" % (digits, x))
    return x


def sanitize(obj, digits=None):
    """JSON-safe copy: non-finite floats become null (strict parsers reject NaN / Infinity); digits: also round the floats."""
    if isinstance(obj, dict):
        return {str(k): sanitize(v, digits) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return [sanitize(v, digits) for v in obj]
    if isinstance(obj, (np.floating,)):
        obj = float(obj)
    if isinstance(obj, (np.integer,)):
        return int(obj)
    if isinstance(obj, (np.bool_,)):
        return bool(obj)
    if isinstance(obj, float):
        if not math.isfinite(obj):
            return None
        return sig(obj, digits) if digits else obj
    return obj


def config_row(r):
    """[us per step (HIP events), fraction of HBM peak, fraction of the fp32 vector peak, launches per step] of a secondary workload."""
    rf = r.get("roofline") or {}
    return [r.get("launch_us_events", r.get("us_per_step_events")), (rf.get("hbm") or {}).get("frac"),
            (rf.get("valu") or {}).get("frac"), rf.get("launches_per_step")]


def build_line(head, also, one_launch, cpu, world, robot_name, use_graph):
    """The contract line as a dict: the driver's keys, `roofline` (scalars + both fractions + the one-launch figure + one
    compact row per secondary workload), `cpu_baseline`, `collective`, `sanity` - and nothing verbose (format_line caps it)."""
    rf = head["roofline"]
    chains = rf.get("launches_per_step", 1)
    by_name = {a.get("workload"): a for a in also if "value" in a}
    configs = {w: config_row(by_name[w]) for w in CONFIG_ROWS if w in by_name}
    toa = {}
    for w in TRAFFIC_ROWS:
        r = (by_name.get(w) or {}).get("roofline") or {}
        if r.get("traffic"):
            toa[w] = r["traffic"] / r["hbm"]["bytes_per_launch"]
    # scalars first: the driver's record keeps the scalar fields of `roofline`, so north_star's claims must be readable off them alone.
    # `frac` follows the contract (HIP events on the launch stream around a region's K launches, median over the repeats);
    # `frac_wall` is the same ratio from `ms_per_step`, the wall-clock mean that `value` is built on (barrier, synchronisation and
    # the amortised statistics reduction included) - always the smaller of the two.
    step_wall_s = head["ms_per_step"] * 1e-3
    hbm_wall = rf["hbm"]["bytes_per_launch"] / step_wall_s / HBM_PEAK
    valu_wall = (rf["valu"]["flops_per_launch"] / step_wall_s / VALU_PEAK) if rf.get("valu") else None

    def row_frac(w, col):
        return configs[w][col] if w in configs else None
    roof = {"bound": rf["bound"], "achieved": rf["achieved"], "peak": rf["peak"], "unit": rf["unit"], "frac": rf["frac"],
            "frac_wall": valu_wall if rf["bound"] == "valu" else hbm_wall,
            "hbm_frac": rf["hbm"]["frac"], "valu_frac": (rf.get("valu") or {}).get("frac"),
            "hbm_frac_wall": hbm_wall, "valu_frac_wall": valu_wall,
            # against what an all-FMA fp32 stream reaches on this chip (109 TFLOP/s: 1.20 ns per wave-instruction and SIMD at 8 waves,
            # profiles/r1_b/microbench_valu.log; packed fp32 moves the same flops per cycle) - the spec peak above is the contract's roof
            "valu_frac_of_measured_fma_rate": (rf.get("valu") or {}).get("frac_of_measured_fma_rate_109TFLOPs"),
            "one_launch_frac": (one_launch or {}).get("frac"), "one_launch_hbm_frac": (one_launch or {}).get("hbm_frac"),
            "euler_262144_hbm_frac": row_frac("msj-262144-euler", 1), "euler_2097152_hbm_frac": row_frac("msj-2097152-euler", 1),
            "fused_env_2097152_hbm_frac": row_frac("fused-env-2097152", 1),
            "traffic_over_algorithmic": (rf["traffic"] / rf["hbm"]["bytes_per_launch"]) if rf.get("traffic") else None,
            "traffic_stale": rf.get("traffic_stale"),
            "traffic": rf["traffic"], "traffic_source": rf.get("traffic_source"), "kernel": head["kernel"],
            "launch_us_events": rf["launch_us_events"], "launches_per_step": chains, "envs_per_launch": rf["envs_per_launch"],
            "hbm": {k: rf["hbm"][k] for k in ("achieved", "peak", "frac", "bytes_per_env_step", "bytes_per_launch")},
            "valu": ({k: rf["valu"][k] for k in ("achieved", "peak", "frac", "flops_per_env_step", "flops_per_launch")}
                     if rf.get("valu") else None),
            "one_launch_us": (one_launch or {}).get("us_events"),
            "one_launch": ({k: one_launch[k] for k in ("us_events", "frac", "hbm_frac")} if one_launch else None),
            "configs_cols": ["us_events", "hbm_frac", "valu_frac", "launches_per_step"], "configs": configs,
            "traffic_over_algorithmic_rows": toa,
            "note": "us_events = HIP events on the launch stream around a region's K steps, median, / K; rocprofv3 in profiles/"}
    cpu_c = None
    if cpu is not None:
        py = cpu.get("python_env_processes") or {}
        cpu_c = {"value": cpu["value"], "unit": cpu["unit"], "cores": cpu["cores"], "kind": cpu["kind"], "sample": cpu["sample"],
                 "value_1_core": cpu.get("value_1_core"), "host_cores_available": cpu.get("host_cores_available"),
                 "by_threads": {th: v["value"] for th, v in (cpu.get("by_threads") or {}).items()},
                 "python_env_processes": ({"value": py.get("value"), "processes": py.get("processes")} if py else None)}
    coll = head["collective"]
    return {
        "metric": "env-steps/sec, MsjRobot (3-DOF/8-tendon) batched rollout" if robot_name == "MsjRobot"
                  else "env-steps/sec, %s batched rollout" % robot_name,
        "value": head["value"], "unit": "env-steps/s",
        "n_gpus": world, "steps": head["steps"], "warmup": head["warmup"], "repeats": head["repeats"],
        "ms_per_step": head["ms_per_step"], "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": head["label"], "robot": robot_name, "envs_per_gpu": head["envs_per_gpu"],
                   "total_envs": head["envs_per_gpu"] * world, "integrator": head["integrator"],
                   "substeps": head["substeps"], "step_size": 0.1,
                   "launch": (("hipGraph replay, one kernel per step" + ("; %d concurrent chains over 1/%d of the batch each" % (chains, chains)
                                                                          if chains > 1 else ""))
                              if use_graph else "eager per-step launches"),
                   "parallelism": ("env shards x%d, RCCL all-reduce of episode statistics every %d steps" % (world, STATS_EVERY))
                                  if world > 1 else "single GPU",
                   "timing": "mean wall time of the event-free repeats of the K-step region (barrier + synchronize both sides, max over ranks); "
                             "roofline from HIP events in alternate repeats (%.0f ms of device time)" % head["timed_device_ms"]},
        "timing_protocol": TIMING_PROTOCOL,
        "ms_per_step_median": head["ms_per_step_median"], "ms_per_step_spread": [head["ms_per_step_min"], head["ms_per_step_max"]],
        "ms_per_step_with_events": head.get("ms_per_step_with_events"), "event_repeats": head.get("event_repeats"),
        "roofline": roof,
        "cpu_baseline": cpu_c,
        "collective": ({k: coll[k] for k in ("backend", "world_size", "allreduce_calls", "payload_bytes", "every_steps",
                                               "us_per_allreduce", "n_env_steps_allreduced", "expected", "ok")} if coll else None),
        "sanity": {"finite": head["finite"], "feasible_frac": head["feasible_frac"], "allreduced_stats": head["stats"]},
    }


def format_line(line):
    """One compact, strictly valid JSON line (no NaN / Infinity tokens, floats to 5 significant digits except the exact counts)."""
    exact = {"value": line.get("value"), "ms_per_step": line.get("ms_per_step")}      # the driver recomputes one from the other
    out = sanitize(line, digits=5)
    for k, v in exact.items():
        if isinstance(v, float) and math.isfinite(v):
            out[k] = v
    if line.get("sanity"):
        out["sanity"]["allreduced_stats"] = sanitize(line["sanity"].get("allreduced_stats"))   # counts: exact
    if line.get("collective"):
        for k in ("n_env_steps_allreduced", "expected"):
            out["collective"][k] = sanitize(line["collective"].get(k))
    return json.dumps(out, separators=(",", ":"), allow_nan=False)


LAUNCH_TIMEOUT_S = float(os.environ.get("ROBOY_BENCH_LAUNCH_TIMEOUT", "1500"))   # the parent of a self-started N-rank run gives up after this


def free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launcher_argv(n, port, argv):
    """The command the driver's contract names for N > 1 (one rank per GPU, rendezvous on 127.0.0.1), around this file."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def self_launch(n, argv, timeout=None):
    """`python bench.py --gpus N` (N > 1) outside a launcher: start the N ranks as a fresh CHILD process tree (never an exec of
    this process, and before this process has made any GPU call - it never makes one), relay rank 0's one JSON line to stdout,
    let the children's stderr through, return the launcher's exit code; on this process's own timeout or SIGTERM / SIGINT the
    whole child process group is killed.  The reference's axis: `SubprocVecEnv` of `num_cpu` workers, train_parallel.py:19-29."""
    import signal
    timeout = LAUNCH_TIMEOUT_S if timeout is None else timeout
    cmd = launcher_argv(n, free_port(), argv)
    sys.stderr.write("bench.py: --gpus %d without a launcher: starting %s\n" % (n, " ".join(cmd)))
    sys.stderr.flush()
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=None, text=True, cwd=ROOT, start_new_session=True)

    def kill_group(sig=signal.SIGKILL):
        try:
            os.killpg(proc.pid, sig)          # start_new_session: the launcher leads its own process group (pgid = its pid)
        except (ProcessLookupError, PermissionError):
            pass

    def on_signal(signum, _frame):
        kill_group(signal.SIGTERM)
        time.sleep(2.0)
        kill_group()
        sys.exit(128 + signum)

    previous = {s: signal.signal(s, on_signal) for s in (signal.SIGTERM, signal.SIGINT)}
    try:
        out, _ = proc.communicate(timeout=timeout)
        rc = proc.returncode
    except subprocess.TimeoutExpired:
        kill_group(signal.SIGTERM)
        try:
            out, _ = proc.communicate(timeout=10)
        except subprocess.TimeoutExpired:
            kill_group()
            out, _ = proc.communicate()
        sys.stderr.write("bench.py: the %d-rank run did not finish within %.0f s: killed\n" % (n, timeout))
        rc = 124
    finally:
        for s, h in previous.items():
            signal.signal(s, h)
        if proc.returncode is None or proc.returncode != 0:
            kill_group()                       # a launcher that failed or was cut short: none of its ranks outlives this call
                                               # (after a clean exit the group is gone and its id is not ours to signal any more)
    lines = [l for l in (out or "").splitlines() if l.strip()]
    contract = [l for l in lines if l.lstrip().startswith("{")]
    for l in lines:                            # anything else a library wrote to the ranks' stdout: not on ours
        if l not in contract:
            sys.stderr.write(l + "\n")
    if contract:
        sys.stdout.write(contract[-1].strip() + "\n")
        sys.stdout.flush()
    elif rc == 0:
        sys.stderr.write("bench.py: the ranks printed no contract line\n")
        rc = 5
    return rc


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args.gpus, sys.argv[1:]))
    # stdout carries exactly one JSON line.  Libraries below write there on their own
    # (RCCL prints a version banner on stdout when its first communicator comes up), so
    # file descriptor 1 points at stderr until the line is ready.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d does not match WORLD_SIZE %d" % (args.gpus, world))
    if args.gpus > 1 and world == 1:
        raise SystemExit("--gpus %d under a launcher that set WORLD_SIZE=1: start bench.py without one (it starts its own ranks) "
                         "or with --nproc-per-node %d" % (args.gpus, args.gpus))
    from gym_roboy_amd.envs.robots import MsjRobot, UpperBodyRobot
    robot = UpperBodyRobot() if args.workload.startswith("upper-body") else MsjRobot()
    # CPU baseline first: its Python env workers are separate processes, started while this
    # process has not initialised the GPU yet (rank 0 at N = 1 only)
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(robot, args.cpu_seconds, args.workload)

    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback for the physics step)")
    # ROBOY_BENCH_BACKEND=gloo is a rehearsal mode for boxes with fewer GPUs than
    # ranks: ranks share the visible GPUs and the tiny collectives go over gloo.
    backend = os.environ.get("ROBOY_BENCH_BACKEND", "nccl")
    local_rank = local_rank % torch.cuda.device_count() if backend == "gloo" else local_rank
    torch.cuda.set_device(local_rank)
    dist = None
    # ROBOY_BENCH_DIST_AT_1=1 with RANK/WORLD_SIZE=1/MASTER_* set drives the whole collective
    # path (RCCL init, statistics all-reduce, barrier, max-reduce) on a one-GPU box; the
    # multi-GPU launches always take it
    if world > 1 or (os.environ.get("ROBOY_BENCH_DIST_AT_1") == "1" and "RANK" in os.environ):
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=backend)

    use_graph = not args.no_graph
    also = []
    # stream capture is not allowed on the legacy default stream: run on a side stream
    with torch.cuda.stream(torch.cuda.Stream()):
        head = run_workload(torch, robot, args.workload, args.steps, args.warmup, args.envs, use_graph,
                            rank, world, dist, args.substeps, args.kernel, args.repeats)
        if world == 1 and not args.no_also:
            for name in ("msj-4096-euler", "msj-262144-euler", "msj-262144-rk4", "msj-2097152-euler", "msj-2097152-rk4",
                         "upper-body-8192-euler", "upper-body-8192-rk4", "upper-body-65536-euler"):
                if name != args.workload:
                    rob = UpperBodyRobot() if name.startswith("upper-body") else MsjRobot()
                    also.append(brief(run_workload(torch, rob, name, None, None, None, use_graph, rank, world, dist)))
            also.append(brief(run_fused_env(torch, MsjRobot(), 2097152)))
            also.append(brief(run_fused_env(torch, UpperBodyRobot(), 8192)))
            also.append(brief(run_fused_env(torch, UpperBodyRobot(), 8192, spread=True)))
            also.append(brief(run_fused_env(torch, UpperBodyRobot(), 65536)))
            also.append(brief(run_fused_env(torch, UpperBodyRobot(), 65536, chains=2)))
            for n_fused in (4096, 2097152):
                also.append(brief(run_fused_rollout(torch, MsjRobot(), n_fused)))
            for fused in (True, False):                      # the consumer, end to end (timesteps/s, not env-steps/s)
                try:
                    also.append(run_ppo_iteration(torch, 65536, fused))
                except Exception as exc:                     # never at the expense of the headline line
                    also.append({"workload": "ppo-65536-%s" % ("fused" if fused else "torch"), "error": repr(exc)[:300],
                                 "finite": False, "feasible_frac": 0.0})

    # the headline workload once more with ONE launch per step over the whole batch (what rb_step_dev / rb_env_step_dev and
    # every per-step caller launch): beside the default form in the same line
    one_launch = None
    if (head["roofline"].get("launches_per_step", 1) > 1 and not os.environ.get("ROBOY_BENCH_NO_ONE_LAUNCH")
            and (world == 1 or args.one_launch)):      # N > 1: on request only (the leg repeats the whole protocol with its barriers on every rank)
        with torch.cuda.stream(torch.cuda.Stream()):
            one = run_workload(torch, robot, args.workload, args.steps, args.warmup, args.envs, use_graph, rank, world, dist,
                               args.substeps, args.kernel, max(3, (head["repeats"] + 3) // 4), force_chains=1)
        one_launch = {"us_events": one["launch_us_events"], "ms_per_step": one["ms_per_step"], "value": one["value"],
                      "frac": one["roofline"]["frac"], "hbm_frac": one["roofline"]["hbm"]["frac"], "repeats": one["repeats"]}

    rc = 0
    if rank == 0:
        line = build_line(head, also, one_launch, cpu, world, type(robot).__name__, use_graph)
        verbose = dict(line, roofline_full=head["roofline"], one_launch=one_launch, cpu_baseline_full=cpu, also=also)
        text = format_line(line)
        # the long form (every secondary workload with both rooflines spelled out) goes beside bench.py and to stderr,
        # never to stdout: the driver keeps a bounded tail of stdout and must find the whole contract line in it
        try:
            with open(os.path.join(ROOT, "bench_also.json"), "w") as fh:
                json.dump(sanitize(verbose), fh, indent=1)
        except OSError as exc:
            sys.stderr.write("bench.py: bench_also.json not written (%s)\n" % exc)
        sys.stderr.write("bench.py: long form follows on stderr\n" + json.dumps(sanitize(verbose)) + "\n")
        sys.stderr.flush()
        sys.stdout.flush()
        os.write(json_fd, (text + "\n").encode())
        if len(text) > LINE_CAP:
            sys.stderr.write("bench.py: the contract line is %d bytes, over the %d-byte cap\n" % (len(text), LINE_CAP))
            rc = 4
        if head["collective"] is not None and not head["collective"]["ok"]:
            sys.stderr.write("bench.py: the all-reduced env-step count %r does not match world x envs x steps = %r\n"
                             % (head["collective"]["n_env_steps_allreduced"], head["collective"]["expected"]))
            rc = 3
    if dist is not None:
        dist.destroy_process_group()
    if rc:
        sys.exit(rc)


if __name__ == "__main__":
    main()
