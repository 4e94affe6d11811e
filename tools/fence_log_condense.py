#!/usr/bin/env python3
"""Condense the two ROCclr packet logs of tools/gpu_fence_probe.sh into profiles/<dir>/chain_fence_scopes.log.
    python tools/fence_log_condense.py gpurun_out/r6_fence profiles/r6_a"""
import re
import sys

src, dst = sys.argv[1], sys.argv[2]
HEAD = """chain_fence_scopes.log - fence scopes of the AQL packets of a two-chain rollout (tools/gpu_fence_probe.sh -> tools/proto/chain_fence_probe.py,
AMD_LOG_LEVEL=4, ROCclr's own packet log; %s, MI355X).  Scopes: 0 = none, 1 = agent, 2 = system.
Workload: MsjRobot 262 144 envs RK4, two chains; phases A / B = 8-step rollouts (graphs captured, then replayed), C = 22 steps
(eager head on both chains, 16-step graphs, join, 2 trailing whole-batch launches), D = read_state (pack kernel behind the join).
Queue q3 = the second chain's stream, q2 = the handle's (caller's) stream.

What the log shows:
  * EVERY kernel dispatch packet - eager launches and graph replays alike - has Header 0xb02: barrier=1, acquire=1, release=1:
    agent-scope acquire at the kernel's start, agent-scope release at its end.  This is what orders one chain's writes before the
    other chain's consumers on the same device; it does not depend on the events' flags.
  * hipEventRecord(chain_join) becomes a BarrierValue marker on the chain's queue: with hipEventDisableSystemFence its header is 0x100
    (acquire=0, release=0: completion only), without the flag 0x1500 (acquire=2, release=2: a system-scope release + acquire, the
    ~3 us the flag saves per event).  hipStreamWaitEvent becomes a BarrierAND on the waiting queue (dep_signal = the marker's
    completion signal) with acquire=0, release=0 in BOTH modes: the wait itself never carried a fence.
  * So with the flag the join is: producer kernel (release=agent) -> marker (completion) -> BarrierAND -> consumer kernel (acquire=agent).
    The system-scope release that is dropped serves the host and other devices only; the host-facing entry points end in
    hipStreamSynchronize, whose own barrier packet (0x1503: acquire=2, release=2) is unchanged (last packet of every phase below).
  * (The fork emitted no packets in this run: the handle's stream was idle, rb_rollout_dev skips the fork event then.)
"""


def condense(fn):
    rows = []
    for l in open(fn):
        if l.startswith("=== PROBE"):
            rows.append(l.rstrip())
            continue
        m = re.search(r":(1083|1267|1360): \d+ us: \[[^\]]*\] SWq=\S+ HWq=\S+ id=(\d+), (.*)", l)
        if not m:
            continue
        body, q = m.group(3), m.group(2)
        if m.group(1) == "1083":
            h = re.search(r"Dispatch Header = (\S+ \([^)]*\))", body).group(1)
            g = re.search(r"grid=\[(\d+)", body).group(1)
            rows.append("  q%s Dispatch %s grid=%s" % (q, h, g))
        else:
            rows.append("  q%s %s" % (q, re.sub(r", rptr=.*", "", body)))
    res, prev, cnt = [], None, 0
    for r in rows + [None]:                     # run-length compress identical consecutive rows
        if r == prev:
            cnt += 1
            continue
        if prev is not None:
            res.append(prev + ("   x%d" % cnt if cnt > 1 else ""))
        prev, cnt = r, 1
    return res


try:
    ver = "HIP " + open(src + "/hip_version.txt").read().split()[0]
except Exception:
    ver = "HIP version not recorded"
out = [HEAD % ver, "packets, chain events fence-free (default build):", ""]
out += condense(src + "/chain_fence_free.log")
out += ["", "packets, ROBOY_SIM_EVENT_SYSTEM_FENCE=1 (events created with hipEventDisableTiming only):", ""]
out += condense(src + "/chain_fence_system.log")
open(dst + "/chain_fence_scopes.log", "w").write("\n".join(out) + "\n")
heads = {}
for mode in ("free", "system"):
    heads[mode] = sorted(set(re.findall(r"Dispatch Header = (\S+ \([^)]*\))", open("%s/chain_fence_%s.log" % (src, mode)).read())))
print("dispatch headers:", heads)
