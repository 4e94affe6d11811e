#!/usr/bin/env python3
"""bench.py - env-steps/s of the batched MsjRobot physics step on N MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload NAME]

One "step" = one lock-step call of the batched ``forward_step_command`` = one
kernel launch that advances every env of the rank's shard by one integrator
step of dt = 0.1.  Actions are i.i.d. U[-1,1) from the Philox
streams, pre-generated as a ring of 4 slabs resident in HBM and rescaled by
0.3 in the kernel.  The default workload is BASELINE.json configs[1]
("MsjRobot 4 096 envs, semi-implicit Euler fp32, 1 MI355X"); the other
single-GPU configs are available through --workload and are also measured
briefly and reported under "also" (they are not the headline).

N > 1: launched by torch.distributed.run, one rank per GPU; envs shard
contiguously (weak scaling: the per-GPU batch is fixed), no data-path
collective; every 100 steps the rank's episode statistics (8 doubles) are
all-reduced over RCCL.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

WORKLOADS = {
    # name: (envs per GPU, integrator, substeps, default steps, default warmup, BASELINE.json config)
    "msj-4096-euler": (4096, "euler", 1, 4000, 200, "configs[1]: MsjRobot 4 096 envs, semi-implicit Euler fp32"),
    "msj-262144-rk4": (262144, "rk4", 1, 400, 40, "configs[2]: MsjRobot 262 144 envs, RK4 fp32"),
    "msj-262144-euler": (262144, "euler", 1, 1000, 100, "configs[4] shard: 262 144 envs per GPU, Euler fp32"),
    "msj-2097152-euler": (2097152, "euler", 1, 300, 30, "large batch: 2 097 152 envs on one GPU, Euler fp32"),
    "upper-body-8192-euler": (8192, "euler", 1, 300, 30, "configs[3]: upper body (20 DOF / 38 tendons) 8 192 envs, Euler fp32"),
    "upper-body-8192-rk4": (8192, "rk4", 1, 100, 10, "configs[3]: upper body (20 DOF / 38 tendons) 8 192 envs, RK4 fp32"),
}
RING = 4
# Steps between two all-reduces of the statistics block: one episode horizon
# (RoboyEnv.max_episode_length = 400 env steps, roboy_env.py:23).
STATS_EVERY = int(os.environ.get("ROBOY_BENCH_STATS_EVERY", "400"))
HBM_PEAK = 8.0e12          # B/s, spec (MI355X_MICROARCH.md "HBM3E peak BW")
HBM_COPY = 6.29e12         # B/s, measured float4 copy (same table)


def pmc_traffic(workload):
    """HBM bytes per launch of the step kernel from the committed rocprofv3 PMC
    passes (profiles/r1_b/hbm_traffic_pmc.json: separate --pmc FETCH_SIZE and
    --pmc WRITE_SIZE runs of this workload, FETCH_SIZE doubled as the gfx950
    guide prescribes).  PMC collection cannot run inside the timed bench, so the
    number is the profiled one, not a live one; None if the workload was not profiled."""
    path = os.path.join(ROOT, "profiles", "r1_b", "hbm_traffic_pmc.json")
    try:
        with open(path) as fh:
            return json.load(fh)[workload]["hbm_bytes_per_launch"]
    except Exception:
        return None


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--workload", default="msj-4096-euler", choices=sorted(WORKLOADS))
    ap.add_argument("--envs", type=int, default=None, help="override envs per GPU")
    ap.add_argument("--substeps", type=int, default=None, help="override integrator substeps per env step")
    ap.add_argument("--kernel", type=int, default=0, help="0 auto, 1 env-per-lane, 2 tendon-per-lane")
    ap.add_argument("--no-graph", action="store_true", help="eager per-step launches instead of hipGraph replay")
    ap.add_argument("--no-also", action="store_true", help="skip the secondary workloads")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    return ap.parse_args()


def run_workload(torch, robot, name, steps, warmup, envs, use_graph, rank, world, dist, substeps=None, kernel=0):
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
    stream = torch.cuda.current_stream()        # main() made a non-default stream current
    sim.set_stream(stream.cuda_stream)          # launches and torch events share one stream
    slab = n_envs * sim.n_t
    ring = torch.empty(RING * slab, dtype=torch.float32, device="cuda")
    for r in range(RING):
        sim.fill_actions_dev(ring.data_ptr() + 4 * r * slab, r)
    # statistics all-reduce: 64 bytes, enqueued IN-LINE on the launch stream (a non-async
    # c10d op runs on the current stream).  Measured on one MI355X (profiles/r1_b/
    # rccl_one_rank_rehearsal.log): letting the collective run concurrently on a second
    # stream (async_op=True, or an own side stream) slows the graph-replayed step kernels
    # from 2.2 to 3.7 us per step for the whole rollout, independent of how often it runs;
    # in line it costs its own ~20 us per call and nothing else.
    stats_ring = [torch.zeros(8, dtype=torch.float64, device="cuda") for _ in range(2)]
    state = {"chunk": 0, "last": stats_ring[0]}
    act_scale = float(robot.get_action_space().high[0])

    def reduce_stats():
        from gym_roboy_amd import _native as nat
        import ctypes
        buf = stats_ring[state["chunk"] % 2]
        state["chunk"] += 1
        nat.check(sim._lib.rb_env_stats_dev(sim.handle, ctypes.c_void_p(buf.data_ptr()), 0))
        if dist.get_backend() == "nccl":
            dist.all_reduce(buf)                                    # RCCL over xGMI
        else:                                                       # rehearsal over gloo: through the host
            host = buf.cpu()
            dist.all_reduce(host)
            buf.copy_(host)
        state["last"] = buf

    def rollout(k):
        done = 0
        while done < k:
            chunk = min(STATS_EVERY, k - done)
            sim.rollout_dev(ring.data_ptr(), RING, chunk, act_scale, use_graph=use_graph)
            done += chunk
            if dist is not None and chunk == STATS_EVERY:
                reduce_stats()

    # 16 untimed steps from the reset state decorrelate the envs, then warm-up
    rollout(16)
    rollout(warmup)
    # untimed rehearsal of the chunk sizes the timed region will use, so that no
    # hipGraph is captured / instantiated inside it (graphs are cached per chunk size)
    rollout(min(steps, STATS_EVERY))
    if steps > STATS_EVERY and steps % STATS_EVERY:
        rollout(steps % STATS_EVERY)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record(stream)
    rollout(steps)
    ev1.record(stream)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    dev_ms = ev0.elapsed_time(ev1)
    if dist is not None:
        t = torch.tensor([wall], dtype=torch.float64, device="cuda" if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall = float(t.item())
    q, qd, feas = sim.read_state()
    info = sim.info()
    sim.close()
    bytes_per_launch = info["bytes_per_env_step"] * n_envs
    launch_s = dev_ms * 1e-3 / steps            # HIP events on the launch stream, per launch
    return {
        "workload": name, "label": label, "envs_per_gpu": n_envs, "integrator": integrator,
        "substeps": nsub, "steps": steps, "warmup": warmup,
        "value": world * n_envs * steps / wall, "ms_per_step": wall * 1e3 / steps,
        "launch_us_events": launch_s * 1e6, "bytes_per_launch": bytes_per_launch,
        "achieved_GBps": bytes_per_launch / launch_s / 1e9,
        "kernel": {1: "msj_step_env_per_lane", 2: "msj_step_tendon_per_lane", 3: "tree_step_wave_per_env"}[info["kernel"]],
        "stats": [float(x) for x in state["last"].cpu()],
        "finite": bool(np.isfinite(q).all() and np.isfinite(qd).all()),
        "feasible_frac": float(feas.mean()),
    }


def run_fused_env(torch, robot, n_envs, steps=300, warmup=30):
    """Secondary line: the fused env layer (rb_env_step_dev: rescale + physics +
    obs/reward/done/goal, 84 + 72 algorithmic bytes per env step), eager launches."""
    from gym_roboy_amd.envs.vec_env import RoboyVecEnv
    env = RoboyVecEnv(robot, n_envs)
    st = torch.cuda.current_stream()
    env.sim.set_stream(st.cuda_stream)
    acts = [torch.rand((n_envs, env.n_t), device="cuda") * 2 - 1 for _ in range(RING)]
    obs = torch.empty((n_envs, 3 * env.n_q), device="cuda")
    rew = torch.empty(n_envs, device="cuda")
    done = torch.empty(n_envs, dtype=torch.int32, device="cuda")
    for t in range(warmup):
        env.step_dev(acts[t % RING].data_ptr(), obs.data_ptr(), rew.data_ptr(), done.data_ptr())
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record(st)
    for t in range(steps):
        env.step_dev(acts[t % RING].data_ptr(), obs.data_ptr(), rew.data_ptr(), done.data_ptr())
    e1.record(st)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    us = e0.elapsed_time(e1) * 1e3 / steps
    bytes_per = 4 * (4 * env.n_q + env.n_t + 1) + 4 * (3 * env.n_q + env.n_q + 6)   # + obs, goal, counter x2, return x2, reward, done
    env.close()
    return {"workload": "fused-env-%d" % n_envs, "label": "fused env layer (RoboyVecEnv.step), %d envs, Euler fp32" % n_envs,
            "value": n_envs * steps / wall, "ms_per_step": wall * 1e3 / steps, "launch_us_events": us,
            "achieved_GBps": n_envs * bytes_per / us / 1e3, "steps": steps, "bytes_per_env_step": bytes_per,
            "frac_of_hbm_peak": n_envs * bytes_per / us / 1e3 * 1e9 / HBM_PEAK}


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
    q, _, _ = sim.read_state()
    sim.close()
    return {"workload": "fused-rollout-%d" % n_envs,
            "label": "open-loop rollout, %d steps fused per launch, %d envs, Euler fp32 (not the per-step contract)"
                     % (steps_per_launch, n_envs),
            "value": n_envs * steps / wall, "ms_per_step": wall * 1e3 / steps, "us_per_step_events": us,
            "steps": steps, "bytes_per_env_step": 4 * sim.n_t, "finite": bool(np.isfinite(q).all())}


def cpu_baseline(robot, seconds, name):
    """The C restatement of the same step (oracle/roboy_oracle.c, fp32 build)
    timed on this box's host cores on a bounded sample of the workload: first
    one thread (the scalar port), then OpenMP over envs on up to 64 cores with
    a sample large enough (>= 256 envs per thread) for the threads to pay off."""
    from oracle.c_oracle import COracle
    from oracle import philox_np as ph
    n_envs, integrator, nsub, *_ = WORKLOADS[name]
    desc = robot.get_description()
    orc = COracle(desc, "f32")
    integ = 0 if integrator == "euler" else 1
    small = desc.n_q <= 3
    cores_avail = len(os.sched_getaffinity(0))
    out = {}
    for threads in sorted({1, min(cores_avail, 64)}):
        n = min(n_envs, 4096 if small else 512) if threads == 1 else min(max(n_envs, 256 * threads), 65536 if small else 16384)
        ids = np.arange(n, dtype=np.uint64)
        slabs = [np.ascontiguousarray(ph.actions(0, ids, r, desc.n_t) * np.float32(0.3)) for r in range(RING)]
        q = np.zeros((n, desc.n_q), np.float32)
        qd = np.zeros((n, desc.n_q), np.float32)
        feas = np.zeros(n, np.uint8)
        for t in range(4):
            orc.step_inplace(q, qd, slabs[t % RING], feas, integrator=integ, n_substeps=nsub, threads=threads)
        budget = seconds / 2
        t0 = time.perf_counter()
        k = 0
        while time.perf_counter() - t0 < budget:
            orc.step_inplace(q, qd, slabs[k % RING], feas, integrator=integ, n_substeps=nsub, threads=threads)
            k += 1
        out[threads] = (n * k / (time.perf_counter() - t0), k, n)
    best = max(out, key=lambda th: out[th][0])
    # the reference's architecture: one Python RoboyEnv per env (train_parallel.py:19-29), one core
    py_loop = None
    if small:
        from gym_roboy_amd.envs import RoboyEnv
        from oracle.cpu_simulation_client import CpuSimulationClient
        import contextlib
        import io
        acts = [a for a in ph.actions(0, np.arange(256, dtype=np.uint64), 0, desc.n_t)]
        # RoboyEnv prints a banner whenever a goal is reached (also while it
        # derives its reward range): keep stdout to the one JSON line
        with contextlib.redirect_stdout(io.StringIO()):
            env = RoboyEnv(simulation_client=CpuSimulationClient(robot))
            env.reset()
            t0 = time.perf_counter()
            k = 0
            while time.perf_counter() - t0 < 2.0:
                if env.step(acts[k % 256])[2]:
                    env.reset()
                k += 1
        py_loop = k / (time.perf_counter() - t0)
    return {
        "value": out[best][0], "unit": "env-steps/s", "cores": best, "kind": "port",
        "sample": "%d envs x %d steps of %s, C fp32 restatement (oracle/roboy_oracle.c), %s"
                  % (out[best][2], out[best][1], name, "OpenMP over envs" if best > 1 else "one thread"),
        "value_1_core": out[1][0], "host_cores_available": cores_avail,
        "python_env_loop_1_core": py_loop,   # RoboyEnv.step over the oracle-backed client, per process
    }


def main():
    args = parse()
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
        raise SystemExit("for --gpus N > 1 launch with: python -m torch.distributed.run --nnodes=1 "
                         "--nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...")
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

    from gym_roboy_amd.envs.robots import MsjRobot, UpperBodyRobot
    robot = UpperBodyRobot() if args.workload.startswith("upper-body") else MsjRobot()
    use_graph = not args.no_graph
    also = []
    # stream capture is not allowed on the legacy default stream: run on a side stream
    with torch.cuda.stream(torch.cuda.Stream()):
        head = run_workload(torch, robot, args.workload, args.steps, args.warmup, args.envs, use_graph,
                            rank, world, dist, args.substeps, args.kernel)
        if world == 1 and not args.no_also:
            for name in ("msj-262144-rk4", "msj-2097152-euler", "upper-body-8192-euler"):
                if name != args.workload:
                    rob = UpperBodyRobot() if name.startswith("upper-body") else MsjRobot()
                    r = run_workload(torch, rob, name, None, None, None, use_graph, rank, world, dist)
                    also.append({k: r[k] for k in ("workload", "label", "value", "ms_per_step", "launch_us_events",
                                                   "achieved_GBps", "steps")} |
                                {"frac_of_hbm_peak": r["achieved_GBps"] * 1e9 / HBM_PEAK})
            also.append(run_fused_env(torch, MsjRobot(), 2097152))
            for n_fused in (4096, 2097152):
                also.append(run_fused_rollout(torch, MsjRobot(), n_fused))
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(robot, args.cpu_seconds, args.workload)

    if rank == 0:
        line = {
            "metric": "env-steps/sec, MsjRobot (3-DOF/8-tendon) batched rollout" if type(robot).__name__ == "MsjRobot"
                      else "env-steps/sec, %s batched rollout" % type(robot).__name__,
            "value": head["value"], "unit": "env-steps/s",
            "n_gpus": world, "steps": head["steps"], "warmup": head["warmup"],
            "ms_per_step": head["ms_per_step"], "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": head["label"], "robot": type(robot).__name__, "envs_per_gpu": head["envs_per_gpu"],
                       "total_envs": head["envs_per_gpu"] * world, "integrator": head["integrator"],
                       "substeps": head["substeps"], "step_size": 0.1,
                       "launch": "hipGraph replay of per-step kernels" if use_graph else "eager per-step launches",
                       "parallelism": "env shards x%d, RCCL all-reduce of episode statistics every %d steps"
                                      % (world, STATS_EVERY) if world > 1 else "single GPU"},
            "roofline": {
                "bound": "hbm", "achieved": head["achieved_GBps"], "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                "frac": head["achieved_GBps"] * 1e9 / HBM_PEAK,
                "traffic": pmc_traffic(args.workload) if args.envs is None and args.substeps is None else None,
                "traffic_source": "rocprofv3 --pmc passes committed in profiles/r1_b/hbm_traffic_pmc.json (bytes per launch)",
                "kernel": head["kernel"], "bytes_per_env_step": head["bytes_per_launch"] // head["envs_per_gpu"],
                "bytes_per_launch": head["bytes_per_launch"], "launch_us_events": head["launch_us_events"],
                "note": "events bracket the whole timed region on the launch stream, so the per-launch "
                        "time includes the kernel boundary; rocprofv3 kernel-only time is in profiles/",
                "launch_floor_us": {"empty_kernel_graph_node": 1.61, "load_store_only_graph_node": 2.16,
                                    "source": "profiles/r1_b/graph_floor.log (tools/graph_floor.hip), 4 096-env grid"}
                                   if head["envs_per_gpu"] == 4096 else None,
            },
            "cpu_baseline": cpu,
            "also": also,
            "sanity": {"finite": head["finite"], "feasible_frac": head["feasible_frac"],
                       "allreduced_stats": head["stats"]},
        }
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(line) + "\n").encode())
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
