"""Does an initialised RCCL communicator change the per-step time of the graph-replayed
rollout on the same GPU?  One process, world size 1, four stages timed with HIP events."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29531")
saved = os.dup(1)
os.dup2(2, 1)          # RCCL prints its banner on stdout

import torch
import torch.distributed as dist
from gym_roboy_amd.envs.robots import MsjRobot
from gym_roboy_amd.envs.simulations import HipBatchSimulation

N, RING, STEPS = 4096, 4, 3200
out = []


def say(*a):
    os.write(saved, (" ".join(str(x) for x in a) + "\n").encode())


with torch.cuda.stream(torch.cuda.Stream()):
    st = torch.cuda.current_stream()
    sim = HipBatchSimulation(MsjRobot(), N)
    sim.set_stream(st.cuda_stream)
    ring = torch.rand((RING, N, 8), device="cuda") * 2 - 1

    def timed(label, graph=True):
        for _ in range(2):
            sim.rollout_dev(ring.data_ptr(), RING, 100, 0.3, use_graph=graph)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record(st)
        for _ in range(STEPS // 100):
            sim.rollout_dev(ring.data_ptr(), RING, 100, 0.3, use_graph=graph)
        e1.record(st)
        torch.cuda.synchronize()
        say("%-44s graph=%d  events %.3f us/step   wall %.3f us/step"
            % (label, graph, e0.elapsed_time(e1) * 1e3 / STEPS, (time.perf_counter() - t0) * 1e6 / STEPS))

    timed("1 before torch.distributed")
    timed("1 before torch.distributed", graph=False)
    if os.environ.get("PROBE_DEVICE_ID") == "1":
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    else:
        dist.init_process_group("nccl", rank=0, world_size=1)
    timed("2 process group up, no communicator yet")
    buf = torch.zeros(8, dtype=torch.float64, device="cuda")
    dist.all_reduce(buf)
    torch.cuda.synchronize()
    timed("3 after the first all_reduce (communicator)")
    timed("3 after the first all_reduce (communicator)", graph=False)
    w = dist.all_reduce(buf, async_op=True)
    w.wait()
    torch.cuda.synchronize()
    timed("3b after an async all_reduce")
    # the bench's chunk loop: statistics kernel and/or async all-reduce every 100 steps
    import ctypes
    from gym_roboy_amd import _native as nat
    bufs = [torch.zeros(8, dtype=torch.float64, device="cuda") for _ in range(2)]

    def chunked(label, do_stats, do_reduce, every=100):
        pending = [None, None]
        sim.rollout_dev(ring.data_ptr(), RING, every, 0.3, use_graph=True)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record(st)
        host_reduce = 0.0
        for c in range(STEPS // every):
            sim.rollout_dev(ring.data_ptr(), RING, every, 0.3, use_graph=True)
            h0 = time.perf_counter()
            slot = c % 2
            if pending[slot] is not None:
                pending[slot].wait()
            if do_stats:
                nat.check(sim._lib.rb_env_stats_dev(sim.handle, ctypes.c_void_p(bufs[slot].data_ptr()), 0))
            if do_reduce:
                pending[slot] = dist.all_reduce(bufs[slot], async_op=True)
            host_reduce += time.perf_counter() - h0
        for w in pending:
            if w is not None:
                w.wait()
        e1.record(st)
        torch.cuda.synchronize()
        say("%-44s every %4d  events %.3f us/step   wall %.3f us/step   host in stats+reduce %.1f us/chunk"
            % (label, every, e0.elapsed_time(e1) * 1e3 / STEPS, (time.perf_counter() - t0) * 1e6 / STEPS,
               host_reduce * 1e6 / (STEPS // every)))

    if os.environ.get("PROBE_BARRIER") == "1":
        dist.barrier()
        torch.cuda.synchronize()
    chunked("5 chunk loop, nothing between chunks", False, False)
    chunked("6 + statistics kernel", True, False)
    chunked("7 + async all_reduce only", False, True)
    chunked("8 + statistics kernel + async all_reduce", True, True)
    chunked("8 + statistics kernel + async all_reduce", True, True, every=800)
    dist.destroy_process_group()
    timed("4 after destroy_process_group")
    sim.close()
