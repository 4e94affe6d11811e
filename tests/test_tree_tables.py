"""Host logic of the joint-tree kernels, checked without a GPU: tree_build() (gym_roboy_amd/csrc/tree_build.hpp,
compiled with g++ through tests/hostmath/tree_tables.cpp) flattens a robot description into the tables the HIP
kernels stage into LDS.  Checked here: every link sits in exactly one (level, octet slot); chains inherit their
parent's slot; exchange slots and child lists are consistent; the per-link gather lists carry every tendon
crossing once with the right sign, sorted and padded; the folded constant lengths + crossing segments reproduce
the oracle's tendon lengths at random poses; the LDS layout has no overlap and fits a CU."""
import ctypes
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class TreeDev(ctypes.Structure):
    _fields_ = ([(n, ctypes.c_int) for n in (
        "n_q", "n_t", "n_cr", "n_levels", "nsub", "n_x", "single_pass", "lw_shift", "q_shift", "ES", "o_W", "o_SQD",
        "o_SPU", "zoff", "o_lc_start", "o_lc_list", "o_lc_link", "o_t_cr_start", "o_rec1", "o_rec5", "o_ext_list",
        "o_tendon", "o_cross", "o_joint")] +
        [("h", ctypes.c_float), ("g", ctypes.c_float * 3)] +
        [(n, ctypes.c_float) for n in ("kps", "pe_k2s", "inv_pe_den", "fv_c1l", "fv_c2l", "fv_c2s", "fv_k")] +
        [("g_words", ctypes.c_void_p), ("n_vec4", ctypes.c_int)])


@pytest.fixture(scope="module")
def lib():
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from build_dir import build_dir
    build = build_dir()
    so = os.path.join(build, "libtreetables.so")
    src = os.path.join(ROOT, "tests", "hostmath", "tree_tables.cpp")
    dep = os.path.join(ROOT, "gym_roboy_amd", "csrc", "tree_build.hpp")
    if not os.path.exists(so) or any(os.path.getmtime(d) > os.path.getmtime(so) for d in (src, dep)):
        subprocess.check_call(["g++", "-O1", "-std=c++17", "-fPIC", "-shared", "-o", so, src])
    return ctypes.CDLL(so)


def build(lib, desc):
    words = np.zeros(1 << 16, np.uint32)
    n = ctypes.c_int(0)
    dev = TreeDev()
    waves = ctypes.c_int(0)
    lds = ctypes.c_long(0)
    rc = lib.tt_build(ctypes.byref(desc.as_c_struct()), ctypes.c_double(0.1), 1, words.ctypes.data_as(ctypes.c_void_p),
                      len(words), ctypes.byref(n), ctypes.byref(dev), ctypes.byref(waves), ctypes.byref(lds))
    assert rc == 0
    consts = (ctypes.c_int * 7)()
    lib.tt_consts(consts)
    k = dict(zip(("E", "LS", "REC1", "REC5", "XSLOT", "TENDON_REC", "CROSS_REC"), consts))
    return words[:n.value].copy(), dev, waves.value, lds.value, k


def robots():
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from gym_roboy_amd.envs.robots import MsjRobot, UpperBodyRobot
    from test_tree_robot_gpu import _hand_robot
    from random_robots import random_tree_robot
    out = {"upper_body": UpperBodyRobot().get_description(), "msj": MsjRobot().get_description(),
           "hand": _hand_robot(MsjRobot())[1]}
    for seed in RANDOM_SEEDS:                      # random topologies: several roots, bushes, tendons that stay on one link
        out["random%d" % seed] = random_tree_robot(seed)[1]
    out["max"] = random_tree_robot(100, n_q=32, n_t=64)[1]                     # the kernel's limits
    out["chain32"] = random_tree_robot(101, n_q=32, n_t=20, shape="chain")[1]    # 32 levels
    out["star32"] = random_tree_robot(102, n_q=32, n_t=24, shape="star")[1]      # a 31-wide level
    return out


RANDOM_SEEDS = (1, 2, 4, 6, 9)
NAMES = ["upper_body", "msj", "hand"] + ["random%d" % s for s in RANDOM_SEEDS] + ["max", "chain32", "star32"]


@pytest.mark.parametrize("name", NAMES)
def test_levels_slots_and_exchange(lib, name):
    desc = robots()[name]
    w, t, waves, lds, k = build(lib, desc)
    wi = w.view(np.int32)
    nq, lw = desc.n_q, 1 << t.lw_shift
    parent = np.asarray(desc.parent)
    level = np.zeros(nq, int)
    for i in range(nq):
        level[i] = 0 if parent[i] < 0 else level[parent[i]] + 1
    assert t.n_levels == level.max() + 1 and t.n_q == nq and t.n_t == desc.n_t
    assert lw >= np.bincount(level).max() and (lw == 1 or lw // 2 < np.bincount(level).max())
    assert bool(t.single_pass) == (k["E"] * lw * 8 <= 64)
    slot, inh, xslot, seen = {}, {}, {}, set()
    for L in range(t.n_levels):
        for x in range(lw):
            r1 = t.o_rec1 + (L * lw + x) * k["REC1"]
            r5 = t.o_rec5 + (L * lw + x) * k["REC5"]
            i = wi[r1]
            assert wi[r5] == i
            if i < 0:
                continue
            assert level[i] == L and i not in seen
            seen.add(i)
            slot[i] = x
            assert wi[r1 + 1] == parent[i] and wi[r5 + 1] == parent[i]
            inh[i] = bool(wi[r1 + 2] & 1)
            np.testing.assert_allclose(w[r1 + 4:r1 + 7].view(np.float32), np.asarray(desc.axis[i], np.float32))
            np.testing.assert_allclose(w[r1 + 7:r1 + 10].view(np.float32), np.asarray(desc.origin[i], np.float32))
            f5 = w[r5 + 12:r5 + 24].view(np.float32)
            assert f5[0] == np.float32(desc.mass[i]) and f5[10] == np.float32(desc.armature[i]) and f5[11] == np.float32(desc.damping[i])
            np.testing.assert_array_equal(f5[1:4], np.asarray(desc.com[i], np.float32))
            np.testing.assert_array_equal(f5[4:10], np.asarray(desc.inertia[i], np.float32))
            massless = desc.mass[i] == 0 and not np.any(np.asarray(desc.inertia[i])[:3])
            assert bool(wi[r5 + 2] & 4) == massless
            if wi[r5 + 2] & 2:
                xslot[i] = wi[r5 + 9]
    assert seen == set(range(nq))
    for i in range(nq):
        if inh[i]:          # the parent sat in the same octet one level up, and no sibling inherits it too
            assert t.single_pass and parent[i] >= 0 and slot[parent[i]] == slot[i]
            assert sum(1 for j in range(nq) if parent[j] == parent[i] and inh[j]) == 1
        assert (i in xslot) == (parent[i] >= 0 and not inh[i])
    assert sorted(xslot.values()) == list(range(t.n_x))
    # per link: REGCHILD flag, inline / listed exchange slots of the other children, ASTORE flag
    for L in range(t.n_levels):
        for x in range(lw):
            r5 = t.o_rec5 + (L * lw + x) * k["REC5"]
            r1 = t.o_rec1 + (L * lw + x) * k["REC1"]
            i = wi[r5]
            if i < 0:
                continue
            kids = [j for j in range(nq) if parent[j] == i]
            ext = sorted(xslot[j] for j in kids if not inh[j])
            n_ext, es = wi[r5 + 3], wi[r5 + 8]
            assert n_ext == len(ext) and bool(wi[r5 + 2] & 1) == any(inh[j] for j in kids)
            listed = [wi[r5 + 4 + q] if q < 4 else wi[t.o_ext_list + es + q] for q in range(n_ext)]
            assert sorted(listed) == ext and sorted(wi[t.o_ext_list + es:t.o_ext_list + es + n_ext]) == ext
            assert bool(wi[r1 + 2] & 2) == (len(ext) > 0)
    # LDS layout of an env's block: links | W (crossing wrenches, zero slot; SQ and the exchange slots alias it) | SQD | SPU
    assert t.o_W == nq * k["LS"] and k["LS"] % 2 == 1 and t.ES % 2 == 1
    assert t.zoff >= 6 * t.n_cr and t.zoff >= nq
    assert t.o_SQD >= t.o_W + max(t.zoff + 6, k["XSLOT"] * t.n_x) and t.o_SPU == t.o_SQD + nq and t.ES >= t.o_SPU + desc.n_t
    assert 1 <= waves <= 8 and lds <= 160 * 1024 and lds == 4 * (len(w) + waves * k["E"] * t.ES)
    assert len(w) % 4 == 0 and t.n_vec4 * 4 == len(w)


@pytest.mark.parametrize("name", NAMES)
def test_crossings_reproduce_the_oracle_geometry_and_gather_lists(lib, name):
    from oracle.physics_np import TendonRobotOracle
    desc = robots()[name]
    w, t, _, _, k = build(lib, desc)
    wi, wf = w.view(np.int32), w.view(np.float32)
    orc = TendonRobotOracle(desc)
    rng = np.random.default_rng(3)
    q = rng.uniform(desc.q_lo, desc.q_hi, (5, desc.n_q))
    length, _ = orc.tendon_geometry(q)
    R, p, _ = orc.kinematics(q)
    m = desc.muscle
    sc = np.sqrt(np.log2(np.e)) / m["fl_width"]
    n_cr = 0
    incid = [[] for _ in range(desc.n_q)]
    for kt in range(desc.n_t):
        c0, c1 = wi[t.o_t_cr_start + kt], wi[t.o_t_cr_start + kt + 1]
        assert c1 >= c0                                      # (a tendon that never leaves one link has no crossing)
        rec = wf[t.o_tendon + kt * k["TENDON_REC"]:][:5]
        l0 = orc.l0[kt]
        np.testing.assert_allclose(rec[0], sc / l0, rtol=1e-6)
        np.testing.assert_allclose(rec[2], m["kp"] * m["setpoint_scale"] / l0, rtol=1e-6)
        np.testing.assert_allclose(rec[3], desc.f_max[kt], rtol=1e-6)
        np.testing.assert_allclose(rec[4], 1.0 / (m["v_max"] * l0), rtol=1e-6)
        lconst = (rec[1] / sc + 1.0) * l0                    # elcs = sc (lconst / l0 - 1)
        total = np.full(len(q), lconst)
        for cr in range(c0, c1):
            o = t.o_cross + cr * k["CROSS_REC"]
            la, lb = wi[o], wi[o + 1]
            ra, rb = wf[o + 2:o + 5].astype(float), wf[o + 5:o + 8].astype(float)
            assert la != lb
            xa = ra if la < 0 else p[:, la] + R[:, la] @ ra
            xb = rb if lb < 0 else p[:, lb] + R[:, lb] @ rb
            total = total + np.linalg.norm(xb - xa, axis=-1)
            if la >= 0:
                incid[la].append((6 * cr) << 1)
            if lb >= 0:
                incid[lb].append(((6 * cr) << 1) | 1)
            n_cr += 1
        np.testing.assert_allclose(total, length[:, kt], rtol=2e-6, atol=2e-7)
    assert n_cr == t.n_cr
    # gather lists: links by falling list length, every list padded to a multiple of 4 with the zero slot
    links = list(wi[t.o_lc_link:t.o_lc_link + desc.n_q])
    assert sorted(links) == list(range(desc.n_q))
    lens = []
    for a, i in enumerate(links):
        s0, s1 = wi[t.o_lc_start + a], wi[t.o_lc_start + a + 1]
        assert s0 % 4 == 0 and (s1 - s0) % 4 == 0
        entries = list(wi[t.o_lc_list + s0:t.o_lc_list + s1])
        real = [e for e in entries if (e >> 1) != t.zoff]
        assert real == incid[i] and len(entries) - len(real) < 4
        assert all((e >> 1) == t.zoff for e in entries[len(real):])
        lens.append(len(real))
    assert lens == sorted(lens, reverse=True)


def test_upper_body_chains_keep_their_octet(lib):
    """The upper body's arms and neck are chains: 17 of its 20 links receive their parent's data in registers,
    2 exchange slots (the shoulders' first joints; the neck's inherits the torso's octet)."""
    desc = robots()["upper_body"]
    w, t, waves, lds, k = build(lib, desc)
    wi = w.view(np.int32)
    lw = 1 << t.lw_shift
    n_inh = sum(1 for q in range(t.n_levels * lw) if wi[t.o_rec1 + q * k["REC1"]] >= 0 and wi[t.o_rec1 + q * k["REC1"] + 2] & 1)
    assert (t.n_levels, lw, t.single_pass, n_inh, t.n_x, t.n_cr) == (10, 4, 1, 17, 2, 38)
    assert waves * (160 * 1024 // lds) >= 16          # 16 waves per CU: 8 192 envs resident at once
