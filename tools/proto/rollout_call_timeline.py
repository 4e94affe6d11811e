#!/usr/bin/env python3
"""Probe (round 6): where a SHORT rb_rollout_dev call spends its time - the driver's K = 20 on the headline batch (262 144 MsjRobot envs,
RK4, two chains) - read off rocprofv3's own clocks: run under `rocprofv3 --kernel-trace --hip-trace`, then tools/proto/rollout_call_timeline.py
--analyse <dir> lines the HIP API records (host) up with the kernel records (device) of every call:

    call            rb_rollout_dev entered (first HIP API record behind the preceding synchronize)
    first kernel    start of the first step kernel (either chain)              -> start latency
    second chain    start of the other chain's first kernel                    -> chain stagger
    tails           end of each chain's last kernel
    sync returns    end of the hipDeviceSynchronize behind the call            -> join + completion + wake-up

usage:  rocprofv3 --kernel-trace --hip-trace --output-format csv -d OUT -- python3 tools/proto/rollout_call_timeline.py [K] [calls]
        python3 tools/proto/rollout_call_timeline.py --analyse OUT"""
import csv
import glob
import os
import statistics
import sys


def run(k, calls):
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    import torch
    from gym_roboy_amd.envs.robots import MsjRobot
    from gym_roboy_amd.envs.simulations import HipBatchSimulation
    n = 262144
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        sim = HipBatchSimulation(MsjRobot(), n, integrator="rk4", seed=1)
        sim.set_stream(st.cuda_stream)
        ring = torch.empty(4 * n * 8, dtype=torch.float32, device="cuda")
        for r in range(4):
            sim.fill_actions_dev(ring.data_ptr() + 4 * r * n * 8, r)
        for _ in range(5):
            sim.rollout_dev(ring.data_ptr(), 4, k, 0.3, use_graph=True)
        torch.cuda.synchronize()
        for _ in range(calls):
            sim.rollout_dev(ring.data_ptr(), 4, k, 0.3, use_graph=True)
            torch.cuda.synchronize()
        sim.close()
    print("ran %d calls of %d steps" % (calls, k))


def analyse(d):
    def load(pattern):
        rows = []
        for f in glob.glob(os.path.join(d, "**", pattern), recursive=True):
            rows += list(csv.DictReader(open(f)))
        return rows
    api = sorted(load("*hip_api_trace.csv"), key=lambda r: int(r["Start_Timestamp"]))
    ker = sorted((r for r in load("*kernel_trace.csv") if "msj_step_env_per_lane" in r["Kernel_Name"]), key=lambda r: int(r["Start_Timestamp"]))
    syncs = [r for r in api if r["Function"] == "hipDeviceSynchronize"]
    out = []
    for a, b in zip(syncs[:-1], syncs[1:]):
        t0, t1 = int(a["End_Timestamp"]), int(b["End_Timestamp"])
        ks = [r for r in ker if t0 <= int(r["Start_Timestamp"]) < t1]
        calls = [r for r in api if t0 <= int(r["Start_Timestamp"]) < int(b["Start_Timestamp"])]
        if len(ks) < 8 or not calls:
            continue
        queues = sorted({r["Queue_Id"] for r in ks})
        if len(queues) != 2:
            continue
        enter = int(calls[0]["Start_Timestamp"])
        first = {q: min(int(r["Start_Timestamp"]) for r in ks if r["Queue_Id"] == q) for q in queues}
        last = {q: max(int(r["End_Timestamp"]) for r in ks if r["Queue_Id"] == q) for q in queues}
        host_done = int(calls[-1]["End_Timestamp"])                 # the last API call of rb_rollout_dev returns
        busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in ks)
        out.append({"steps": len(ks) // 2, "first_kernel": (min(first.values()) - enter) / 1e3, "stagger": (max(first.values()) - min(first.values())) / 1e3,
                    "host_enqueue": (host_done - enter) / 1e3, "tails_apart": abs(last[queues[0]] - last[queues[1]]) / 1e3,
                    "device_span": (max(last.values()) - min(first.values())) / 1e3, "after_last_kernel": (t1 - max(last.values())) / 1e3,
                    "sync_call": (t1 - int(b["Start_Timestamp"])) / 1e3, "total": (t1 - enter) / 1e3, "kernel_us_sum": busy / 1e3,
                    "api_calls": len(calls)})
    if not out:
        print("no two-chain calls found in", d)
        return
    print("%d calls of %d steps (two chains); medians, us:" % (len(out), out[0]["steps"]))
    for key, what in (("total", "call entered -> hipDeviceSynchronize returns"), ("first_kernel", "call entered -> first kernel starts"),
                      ("stagger", "first kernel of one chain -> first kernel of the other"), ("host_enqueue", "call entered -> the call's last HIP API returns (host side)"),
                      ("device_span", "first kernel starts -> last kernel ends"), ("tails_apart", "the chains' last kernels end this far apart"),
                      ("after_last_kernel", "last kernel ends -> hipDeviceSynchronize returns (join, completion signal, host wake-up)"),
                      ("sync_call", "duration of the hipDeviceSynchronize call itself"), ("kernel_us_sum", "sum of the kernels' durations (both chains)"),
                      ("api_calls", "HIP API calls per rb_rollout_dev call")):
        v = [o[key] for o in out]
        print("  %-18s %8.2f   (min %.2f, max %.2f)   %s" % (key, statistics.median(v), min(v), max(v), what))
    st = out[0]["steps"]
    print("  per step: device span %.2f us, total %.2f us; not covered by the device span: %.1f us per call"
          % (statistics.median(o["device_span"] for o in out) / st, statistics.median(o["total"] for o in out) / st,
             statistics.median(o["total"] - o["device_span"] for o in out)))


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--analyse":
        analyse(sys.argv[2])
    else:
        run(int(sys.argv[1]) if len(sys.argv) > 1 else 20, int(sys.argv[2]) if len(sys.argv) > 2 else 40)
