"""Per-step time of the graph-replayed rollout when the launch stream is created before /
after the RCCL communicator (world size 1).  PROBE_ORDER=early|late, one process each."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29551")
saved = os.dup(1)
os.dup2(2, 1)

import torch
import torch.distributed as dist
from gym_roboy_amd.envs.robots import MsjRobot
from gym_roboy_amd.envs.simulations import HipBatchSimulation

N, RING, STEPS = 4096, 4, 3200
order = os.environ.get("PROBE_ORDER", "late")
n_dummy = int(os.environ.get("PROBE_DUMMY_STREAMS", "0"))


def comm_up():
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    buf = torch.zeros(8, dtype=torch.float64, device="cuda")
    dist.all_reduce(buf)
    torch.cuda.synchronize()


torch.cuda.set_device(0)
if order == "late":
    comm_up()
dummies = [torch.cuda.Stream() for _ in range(n_dummy)]
side = torch.cuda.Stream()
if order == "early":
    comm_up()
with torch.cuda.stream(side):
    st = torch.cuda.current_stream()
    sim = HipBatchSimulation(MsjRobot(), N)
    sim.set_stream(st.cuda_stream)
    ring = torch.rand((RING, N, 8), device="cuda") * 2 - 1
    for _ in range(3):
        sim.rollout_dev(ring.data_ptr(), RING, 100, 0.3, use_graph=True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    bufs = [torch.zeros(8, dtype=torch.float64, device="cuda") for _ in range(2)]
    pend = [None, None]
    mode = os.environ.get("PROBE_REDUCE", "none")
    e0.record(st)
    for c in range(STEPS // 100):
        sim.rollout_dev(ring.data_ptr(), RING, 100, 0.3, use_graph=True)
        if mode == "async":
            if pend[c % 2] is not None:
                pend[c % 2].wait()
            pend[c % 2] = dist.all_reduce(bufs[c % 2], async_op=True)
        elif mode == "sync":
            dist.all_reduce(bufs[c % 2])
    e1.record(st)
    torch.cuda.synchronize()
    os.write(saved, ("stream created %-5s (+%d dummy streams first) reduce=%s: %.3f us/step\n"
                     % (order, n_dummy, os.environ.get("PROBE_REDUCE", "none"), e0.elapsed_time(e1) * 1e3 / STEPS)).encode())
    sim.close()
dist.destroy_process_group()
