#!/usr/bin/env python3
"""Could the RK4 stages of a joint tree overlap?  (round-4 verdict, item 4; NEXT.md candidate 3)

Stage s + 1 of the saturated RK4 of tree_lane.hpp evaluates the acceleration at q0 + c * kq_s with kq_s = sat(v0 + c * kv_(s-1)):
its POSITIONS are known as soon as stage s - 1 has finished, so everything in the acceleration that depends on q alone (link
frames, tendon geometry, articulated inertias and what the backward pass derives from them) could run in another wave BESIDE
stage s's velocity / force half - if its results can be handed over.  This tool reads a generated acceleration
(gym_roboy_amd/csrc/tree_lane_baked.hpp: straight-line single-assignment statements), classifies every statement by what it
depends on (q | qd | spu), and counts

  * the statements of the position-only half and of the rest (issue slots: a pair statement is one; flops: two),
  * the values that cross from the position-only half into the rest (registers: a pair value is two) - the hand-over,
  * per link group as the text orders them.

No GPU; the text is the one the library compiles.

    python tools/stage_overlap_count.py [generated header] [--json out.json]
"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
Q, QD, SPU = 1, 2, 4
TMP = re.compile(r"(?<![A-Za-z0-9_])t(\d+)(?![A-Za-z0-9_])")
LDS = re.compile(r"RBL_LDS\((\d+)\)")


def analyse(path):
    text = open(path).read()
    start = text.index("RBL_FN void rbl_accel(")
    body = text[text.index("{\n", start) + 2:]
    body = body[:body.index("\n}\n")]
    dep, pair, lds_dep, lds_pair = {}, {}, {}, {}
    stmts = []          # (target or None, mask, is_pair, names used, text)
    for line in body.splitlines():
        s = line.strip()
        if not s or s.startswith("//") or s.startswith("RBL_SCHED_BARRIER"):
            continue
        m = re.match(r"const (float|rbl_f2) t(\d+) = (.*);$", s)
        if m:
            kind, tgt, expr = m.group(1), int(m.group(2)), m.group(3)
            lhs = None
        else:
            m2 = re.match(r"(RBL_LDS\(\d+\)|qdd\[\d+\]) = (.*);$", s)
            if not m2:
                raise SystemExit("unparsed statement: " + s)
            kind, tgt, lhs, expr = None, None, m2.group(1), m2.group(2)
        mask = 0
        used = [int(x) for x in TMP.findall(expr)]
        for u in used:
            mask |= dep[u]
        if re.search(r"(?<![a-z])q\[\d+\]", expr):
            mask |= Q
        if "qd[" in expr:
            mask |= QD
        if "spu[" in expr:
            mask |= SPU
        lds_used = [int(x) for x in LDS.findall(expr)]
        for k in lds_used:
            mask |= lds_dep[k]
        if tgt is not None:
            dep[tgt] = mask
            pair[tgt] = kind == "rbl_f2"
        elif lhs.startswith("RBL_LDS"):
            k = int(LDS.match(lhs).group(1))
            lds_dep[k] = mask
        stmts.append((tgt, lhs, mask, kind == "rbl_f2", used, lds_used, s))
    pos_only = lambda m: (m & ~Q) == 0
    n_pos = n_rest = f_pos = f_rest = 0
    crossing, crossing_lds = set(), set()
    for tgt, lhs, mask, pr, used, lds_used, s in stmts:
        w = (2 if pr else 1) * (2 if "rbl_fma(" in s else 1)
        arith = tgt is not None and "RBL_LDS(" not in s
        if pos_only(mask):
            n_pos += 1
            f_pos += w if arith else 0
        else:
            n_rest += 1
            f_rest += w if arith else 0
            for u in used:
                if pos_only(dep[u]):
                    crossing.add(u)
            for k in lds_used:
                if pos_only(lds_dep[k]):
                    crossing_lds.add(k)
    cross_regs = sum(2 if pair[u] else 1 for u in crossing)
    # of the crossing values: those that are literal-cheap to recompute (one statement from q-only leaves) do not shrink the
    # count much - list the distribution by how many statements their own cone has
    cone = {}
    for tgt, lhs, mask, pr, used, lds_used, s in stmts:
        if tgt is not None:
            cone[tgt] = 1 + sum(cone[u] for u in used)
    small = sum((2 if pair[u] else 1) for u in crossing if cone[u] <= 3)
    curve = tradeoff(stmts, dep, pair, pos_only) if TRADEOFF else None
    return {"header": os.path.relpath(path, ROOT), "statements": n_pos + n_rest, "tradeoff": curve,
            "position_only": {"statements": n_pos, "flops": f_pos}, "velocity_force_half": {"statements": n_rest, "flops": f_rest},
            "handover": {"temporaries": len(crossing), "registers": cross_regs, "lds_slots_already_parked": len(crossing_lds),
                         "registers_with_a_cone_of_3_statements_or_fewer": small},
            "handover_bytes_per_64_envs": (cross_regs + len(crossing_lds)) * 64 * 4}


TRADEOFF = True


def tradeoff(stmts, dep, pair, pos_only):
    """The best partial hand-over: a predecessor-closed set S of position-only statements runs in the other wave, the main wave
    recomputes the rest; maximise (statements moved) - lam * (registers handed over) by a minimum cut, for a sweep of lam.
    Returns rows [lam, statements moved, registers handed over, KB per 64 envs]."""
    import networkx as nx
    INF = 10 ** 9
    nodes = [t for (t, lhs, m, pr, used, lu, s) in stmts if t is not None and pos_only(m) and "RBL_LDS(" not in s]
    nodeset = set(nodes)
    succ = {t: [] for t in nodes}
    to_rest = {t: False for t in nodes}
    for (t, lhs, m, pr, used, lu, s) in stmts:
        inside = t is not None and t in nodeset
        for u in used:
            if u in nodeset:
                if inside:
                    succ[u].append(t)
                else:
                    to_rest[u] = True
    rows = []
    for lam in (1, 2, 2.5, 3, 3.5, 4, 5, 6, 8):
        g = nx.DiGraph()
        for v in nodes:
            g.add_edge("s", ("x", v), capacity=100)                                   # not moved: one statement stays (x 100: integer capacities)
            w = 2 if pair[v] else 1
            g.add_edge(("x", v), ("g", v), capacity=int(100 * lam * w))                 # handed over: lam per register
            for u in succ[v]:
                g.add_edge(("g", v), ("x", u), capacity=INF)
                g.add_edge(("x", u), ("x", v), capacity=INF)                          # closure: u moved => v moved
            if to_rest[v]:
                g.add_edge(("g", v), "t", capacity=INF)
        g.add_node("t")
        _, (src, _snk) = nx.minimum_cut(g, "s", "t")
        moved = [v for v in nodes if ("x", v) in src]
        ms = set(moved)
        regs = sum((2 if pair[v] else 1) for v in moved if to_rest[v] or any(u not in ms for u in succ[v]))
        rows.append([lam, len(moved), regs, round(regs * 256 / 1024.0, 1)])
    return rows


if __name__ == "__main__":
    argv = list(sys.argv[1:])
    out = None
    if "--json" in argv:
        out = argv[argv.index("--json") + 1]
        del argv[argv.index("--json"):argv.index("--json") + 2]
    args = [a for a in argv if not a.startswith("--")]
    path = args[0] if args else os.path.join(ROOT, "gym_roboy_amd", "csrc", "tree_lane_baked.hpp")
    r = analyse(path)
    print(json.dumps(r, indent=1))
    if out:
        with open(out, "w") as fh:
            json.dump(r, fh, indent=1)
