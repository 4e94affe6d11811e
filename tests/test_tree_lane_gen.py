"""The source generator of the env-per-lane joint-tree kernels (gym_roboy_amd/csrc/tree_lane_gen.hpp), checked
without a GPU: the text it writes for a robot is compiled with g++ (tests/hostmath/tree_lane_host.cpp supplies the
host forms of the few device primitives) and its fp32 accelerations are compared with the fp64 oracle
(oracle/physics_np.py: Jacobian-sum mass matrix, RNE bias, dense solve - an algorithm independent of the generated
articulated-body code) on the upper body and on seeded random robots: random topologies, arbitrary axes, massless
links, tendons over the base, over several links and inside one link.  Also: the committed tree_lane_baked.hpp is
what the generator writes today for the committed upper body."""
import ctypes
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from build_dir import build_dir  # noqa: E402
BUILD = build_dir()


def host_accel(desc, tag, lds_c=True, pack=True):
    """generate -> g++ -> ctypes; returns (accel(q, qd, sp) -> qdd, (lds_slots, statements))."""
    import gen_tree_lane_baked as gen
    os.makedirs(BUILD, exist_ok=True)
    hdr = os.path.join(BUILD, "lane_%s.hpp" % tag)
    info = gen.generate(desc, hdr, lds_c, pack)
    so = os.path.join(BUILD, "liblane_%s.so" % tag)
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", '-DRBL_GENERATED="%s"' % hdr,
                           "-o", so, os.path.join(ROOT, "tests", "hostmath", "tree_lane_host.cpp")])
    lib = ctypes.CDLL(so)
    dims = (ctypes.c_int * 3)()
    lib.tl_dims(dims)
    assert (dims[0], dims[1]) == (desc.n_q, desc.n_t) and dims[2] == info[0]

    def accel(q, qd, sp):
        q = np.ascontiguousarray(q, np.float32); qd = np.ascontiguousarray(qd, np.float32)
        sp = np.ascontiguousarray(sp, np.float32)
        out = np.zeros_like(q)
        p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
        assert lib.tl_accel(p(q), p(qd), p(sp), p(out), q.shape[0]) == 0
        return out
    return accel, info


def check(desc, tag, n=24, tol=2e-4, lds_c=True, pack=True):
    from oracle.physics_np import TendonRobotOracle
    accel, info = host_accel(desc, tag, lds_c, pack)
    rng = np.random.default_rng(7)
    q = rng.uniform(0.9 * desc.q_lo, 0.9 * desc.q_hi, (n, desc.n_q)).astype(np.float32)
    qd = rng.uniform(-desc.qd_max, desc.qd_max, (n, desc.n_q)).astype(np.float32)
    sp = rng.uniform(-0.3, 0.3, (n, desc.n_t)).astype(np.float32)
    ref = TendonRobotOracle(desc).acceleration(q.astype(np.float64), qd.astype(np.float64), sp.astype(np.float64))
    got = accel(q, qd, sp)
    # fp32 vs fp64: relative to the size of the accelerations of the env (deep chains with light links have large ones)
    scale = np.maximum(1.0, np.abs(ref).max(axis=1, keepdims=True))
    err = np.abs(got - ref) / scale
    assert np.isfinite(got).all()
    assert err.max() < tol, "generated acceleration differs from the oracle: %g" % err.max()
    return info


def test_upper_body_generated_acceleration_matches_oracle():
    from gym_roboy_amd.envs.robots import UpperBodyRobot
    slots, stmts, _, flops, live = check(UpperBodyRobot().get_description(), "upper_body")
    assert live < 400                           # the emission order keeps the live set inside a SIMD's register file
    assert slots <= 120 and stmts < 10000       # fits four waves' LDS regions on a CU; the folding still works
    # the executed-flop figure bench.py prices the lane kernel with (profiles/flops_per_env_step.json) is this count
    import json
    rec = json.load(open(os.path.join(ROOT, "profiles", "flops_per_env_step.json")))["UpperBodyRobot/euler/lane"]
    assert rec["accel_flops"] == flops


def test_upper_body_without_lds_parking_is_the_same_function():
    from gym_roboy_amd.envs.robots import UpperBodyRobot
    check(UpperBodyRobot().get_description(), "upper_body_nolds", lds_c=False)


def test_the_two_arms_of_the_upper_body_are_written_as_one_stream_of_pair_values():
    """Mates: the arms (7 links, 13 tendons each) pair up; the statements drop by a third against the unpaired text, which
    computes the same function."""
    import gen_tree_lane_baked as gen
    from gym_roboy_amd.envs.robots import UpperBodyRobot
    desc = UpperBodyRobot().get_description()
    mate, tmate = gen.mates(desc)
    assert mate[6:13] == list(range(13, 20)) and mate[13:20] == [-2] * 7 and mate[:6] == [-1] * 6
    assert tmate[12:25] == list(range(25, 38)) and tmate[25:38] == [-2] * 13 and tmate[:12] == [-1] * 12
    _, stmts_plain, _, flops_plain, _ = check(desc, "upper_body_unpaired", pack=False)
    hdr = os.path.join(BUILD, "lane_upper_body_paired.hpp")
    _, stmts, _, flops, _ = gen.generate(desc, hdr)
    assert stmts < 0.68 * stmts_plain and abs(flops - flops_plain) < 0.02 * flops_plain
    text = open(hdr).read()
    assert text.count("const rbl_f2 ") > 2500 and "RBL_K2(" in text and "rbl_hsum(" in text and "rbl_fma(" in text
    assert "rbl_f2" not in open(os.path.join(BUILD, "lane_upper_body_unpaired.hpp")).read()


@pytest.mark.parametrize("seed", [0, 1, 2, 3])
def test_robots_with_two_structurally_identical_branches(seed):
    """Random trunk + two branches of the same structure and different constants (+ a third, plain branch): paired text and
    unpaired text against the oracle; a tendon from one branch to the other forbids the pairing."""
    import gen_tree_lane_baked as gen
    from gym_roboy_amd.envs.robots.description import RobotDescription
    from random_robots import random_mirrored_spec
    nb = 3 + seed
    desc = RobotDescription(random_mirrored_spec(seed, n_branch=nb, n_t_branch=3 + seed))
    mate, tmate = gen.mates(desc)
    assert mate[2:2 + nb] == list(range(2 + nb, 2 + 2 * nb)) and mate[2 + nb:2 + 2 * nb] == [-2] * nb and all(m == -1 for m in mate[2 + 2 * nb:])
    assert sum(1 for m in tmate if m >= 0) == 3 + seed == sum(1 for m in tmate if m == -2)
    check(desc, "mirrored%d" % seed, tol=5e-4)
    check(desc, "mirrored%d_plain" % seed, tol=5e-4, pack=False)
    crossed = RobotDescription(random_mirrored_spec(seed, n_branch=nb, n_t_branch=3 + seed, extra_cross=True))
    mate, tmate = gen.mates(crossed)
    assert all(m == -1 for m in mate) and all(m == -1 for m in tmate)
    check(crossed, "mirrored%d_crossed" % seed, tol=5e-4)


def test_msj_as_a_joint_tree():
    from gym_roboy_amd.envs.robots import MsjRobot
    check(MsjRobot().get_description(), "msj")


@pytest.mark.parametrize("seed", [0, 1, 2, 3, 4, 5, 6, 7])
def test_random_robots(seed):
    from gym_roboy_amd.envs.robots.description import RobotDescription
    from random_robots import random_tree_spec
    desc = RobotDescription(random_tree_spec(seed))
    check(desc, "random%d" % seed, tol=5e-4)


@pytest.mark.parametrize("shape,n_q", [("chain", 17), ("star", 12), ("chain", 32)])
def test_random_shapes(shape, n_q):
    from gym_roboy_amd.envs.robots.description import RobotDescription
    from random_robots import random_tree_spec
    desc = RobotDescription(random_tree_spec(40 + n_q, n_q=n_q, shape=shape))
    check(desc, "shape_%s%d" % (shape, n_q), tol=2e-3 if n_q > 20 else 5e-4)


def test_in_tree_generated_headers_are_current():
    """`make` writes the upper body's generated headers next to the kernels (they are not committed); if they are there,
    they must be what the generator writes today - a stale header would be compiled into the library and its hash would
    no longer match the text the library regenerates at run time (the handle would silently fall back to other kernels)."""
    import gen_tree_lane_baked as gen
    from gym_roboy_amd.envs.robots import UpperBodyRobot
    csrc = os.path.join(ROOT, "gym_roboy_amd", "csrc")
    if not os.path.exists(os.path.join(csrc, "tree_lane_baked.hpp")):
        pytest.skip("the library has not been built here")
    fresh = os.path.join(BUILD, "tree_lane_baked_fresh.hpp")
    gen.generate(UpperBodyRobot().get_description(), fresh)
    assert open(fresh).read() == open(os.path.join(csrc, "tree_lane_baked.hpp")).read(), "run make -C gym_roboy_amd/csrc"
    gen.generate_split(UpperBodyRobot().get_description(), fresh, max_helpers=gen.SPLIT_HELPERS, helper_share=gen.SPLIT_HELPER_SHARE, two_sweeps=gen.SPLIT_TWO_SWEEPS, cut=gen.SPLIT_CUT, share_trunk=gen.SPLIT_SHARE_TRUNK)
    assert open(fresh).read() == open(os.path.join(csrc, "tree_lane_split_baked.hpp")).read(), "run make -C gym_roboy_amd/csrc"


def test_generated_text_does_not_depend_on_the_host_compiler():
    """The library (hipcc's host compiler) regenerates the text at run time and compares its hash with the committed
    header's (written through g++): argument-evaluation order must not leak into the text."""
    import ctypes
    import shutil
    import gen_tree_lane_baked as gen
    from gym_roboy_amd.envs.robots import RobotDescription, UpperBodyRobot
    from random_robots import random_tree_spec
    clang = "/opt/rocm/lib/llvm/bin/clang++"
    if not os.path.exists(clang):
        clang = shutil.which("clang++")
    if not clang:
        pytest.skip("no clang++ to compare g++ with")
    so = os.path.join(BUILD, "libgen_tree_lane_clang.so")
    subprocess.check_call([clang, "-O2", "-std=c++17", "-fPIC", "-shared", "-o", so,
                           os.path.join(ROOT, "gym_roboy_amd", "csrc", "gen_tree_lane.cpp")])
    lib = ctypes.CDLL(so)
    for k, desc in enumerate([UpperBodyRobot().get_description()] + [RobotDescription(random_tree_spec(s)) for s in (2, 3)]):
        a = os.path.join(BUILD, "cmp_gcc_%d.hpp" % k); b = os.path.join(BUILD, "cmp_clang_%d.hpp" % k)
        gen.generate(desc, a)
        assert lib.rb_gen_tree_lane(ctypes.byref(desc.as_c_struct()), 1, b.encode(), None, None, None, None, None) == 0
        assert open(a).read() == open(b).read()


def _hiprtc_compile(src, name, exprs):
    """Compile `src` for gfx950 with hiprtc (no GPU needed) with the options rbj::compile_and_load uses; returns the log on failure."""
    import ctypes
    try:
        rtc = ctypes.CDLL("libhiprtc.so")
    except OSError:
        try:
            rtc = ctypes.CDLL("/opt/rocm/lib/libhiprtc.so")
        except OSError:
            pytest.skip("hiprtc is not installed")
    prog = ctypes.c_void_p()
    assert rtc.hiprtcCreateProgram(ctypes.byref(prog), src.encode(), name.encode(), 0, None, None) == 0
    for e in exprs:
        assert rtc.hiprtcAddNameExpression(prog, e.encode()) == 0
    opts = [b"--offload-arch=gfx950", b"-O3", b"-std=c++17", b"-fno-slp-vectorize",
            ("-I" + os.path.join(ROOT, "gym_roboy_amd", "csrc")).encode()]
    arr = (ctypes.c_char_p * len(opts))(*opts)
    rc = rtc.hiprtcCompileProgram(prog, len(opts), arr)
    log = ""
    if rc:
        n = ctypes.c_size_t(0)
        rtc.hiprtcGetProgramLogSize(prog, ctypes.byref(n))
        buf = ctypes.create_string_buffer(n.value + 1)
        rtc.hiprtcGetProgramLog(prog, buf)
        log = buf.value.decode(errors="replace")
    size = ctypes.c_size_t(0)
    if not rc:
        rtc.hiprtcGetCodeSize(prog, ctypes.byref(size))
    rtc.hiprtcDestroyProgram(ctypes.byref(prog))
    return rc, log, size.value


def test_hiprtc_builds_the_kernels_of_a_random_robot():
    """What rblj::build (csrc/tree_lane_jit.hpp) hands to hiprtc at run time, compiled here for gfx950: the run-time
    compiler has no standard headers, so every name the kernels use must come from rtc_compat.hpp."""
    import gen_tree_lane_baked as gen
    from gym_roboy_amd.envs.robots import RobotDescription
    from random_robots import random_tree_spec
    hdr = os.path.join(BUILD, "lane_rtc.hpp")
    gen.generate(RobotDescription(random_tree_spec(3)), hdr)
    text = open(hdr).read()
    text = text[:text.rindex("#define RBL_TEXT_HASH")]
    src = '#include "tree_lane_defs.hpp"\n#define RBL_NS rbl_jit\n' + text + '#include "tree_lane.hpp"\n'
    for kern in ("rbl_jit::tree_lane_step<1>", "rbl_jit::tree_lane_env_step<0>"):
        rc, log, size = _hiprtc_compile(src, "roboy_tree_lane_jit.hip", [kern])
        assert rc == 0, log[:2000]
        assert size > 10000


def test_hiprtc_builds_the_lean_split_kernels_of_a_random_robot():
    """What build_split2_kernel (csrc/roboy_sim.hip) hands to hiprtc for rb_select_kernel(6) on a robot without ahead-of-time
    instances: the two-part split text behind `#define RBL_LEAN 1` - parking slots in registers, exchange area over the row
    image (tree_lane_split.hpp) - compiled here for gfx950."""
    import gen_tree_lane_baked as gen
    from gym_roboy_amd.envs.robots import RobotDescription
    from random_robots import random_tree_spec
    hdr = os.path.join(BUILD, "lane_split2_rtc.hpp")
    info = gen.generate_split(RobotDescription(random_tree_spec(5)), hdr, max_parts=2, share_trunk=1)
    assert info["n_parts"] == 2 and info["n_helpers"] == 0
    text = open(hdr).read()
    text = text[:text.rindex("#define RBL_SPLIT_TEXT_HASH")]
    src = '#include "tree_lane_defs.hpp"\n#define RBL_NS rbl_jit_split2\n#define RBL_LEAN 1\n' + text + '#include "tree_lane_split.hpp"\n'
    for kern in ("rbl_jit_split2::tree_split_step<1>", "rbl_jit_split2::tree_split_env_step<0>"):
        rc, log, size = _hiprtc_compile(src, "roboy_tree_split2_jit.hip", [kern])
        assert rc == 0, log[:2000]
        assert size > 10000


def host_split_accel(desc, tag, max_parts=4, max_helpers=0):
    """generate_split -> g++ -> ctypes; returns (accel(q, qd, sp) -> (qdd, trunk mismatches), info).  Every part - and every
    helper wave - runs in a thread of its own; the barriers of the text are pthread barriers."""
    import gen_tree_lane_baked as gen
    os.makedirs(BUILD, exist_ok=True)
    hdr = os.path.join(BUILD, "lane_split_%s.hpp" % tag)
    info = gen.generate_split(desc, hdr, max_parts, max_helpers)
    so = os.path.join(BUILD, "liblane_split_%s.so" % tag)
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-fPIC", "-shared", "-pthread", "-ffp-contract=off", '-DRBL_GENERATED="%s"' % hdr,
                           "-o", so, os.path.join(ROOT, "tests", "hostmath", "tree_lane_split_host.cpp")])
    lib = ctypes.CDLL(so)
    dims = (ctypes.c_int * 6)()
    lib.tl_dims(dims)
    assert (dims[0], dims[1], dims[2], dims[5]) == (desc.n_q, desc.n_t, info["n_parts"], info["n_helpers"])

    def accel(q, qd, sp):
        q = np.ascontiguousarray(q, np.float32); qd = np.ascontiguousarray(qd, np.float32)
        sp = np.ascontiguousarray(sp, np.float32)
        out = np.zeros_like(q)
        p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
        return out, lib.tl_accel(p(q), p(qd), p(sp), p(out), q.shape[0])
    return accel, info


def check_split(desc, tag, n=12, tol=2e-4, max_parts=4, max_helpers=0):
    from oracle.physics_np import TendonRobotOracle
    accel, info = host_split_accel(desc, tag, max_parts, max_helpers)
    rng = np.random.default_rng(11)
    q = rng.uniform(0.9 * desc.q_lo, 0.9 * desc.q_hi, (n, desc.n_q)).astype(np.float32)
    qd = rng.uniform(-desc.qd_max, desc.qd_max, (n, desc.n_q)).astype(np.float32)
    sp = rng.uniform(-0.3, 0.3, (n, desc.n_t)).astype(np.float32)
    ref = TendonRobotOracle(desc).acceleration(q.astype(np.float64), qd.astype(np.float64), sp.astype(np.float64))
    got, mismatches = accel(q, qd, sp)
    assert mismatches == 0, "the trunk's accelerations differ between the parts"
    scale = np.maximum(1.0, np.abs(ref).max(axis=1, keepdims=True))
    err = np.abs(got - ref) / scale
    assert np.isfinite(got).all() and err.max() < tol, err.max()
    return info


def test_upper_body_split_form_matches_oracle():
    """Several waves per env group: trunk (spine) in every part, the arms and the neck in parts of their own."""
    from gym_roboy_amd.envs.robots import UpperBodyRobot
    info = check_split(UpperBodyRobot().get_description(), "upper_body")
    assert info["n_parts"] == 3 and info["part_of_joint"][:3] == [-1, -1, -1]
    assert len({info["part_of_joint"][j] for j in range(6, 13)}) == 1 and len({info["part_of_joint"][j] for j in range(13, 20)}) == 1
    assert info["max_stmt"] < 0.5 * 8103                      # a step waits for less than half the one-wave stream (8 103 statements unpaired)


def test_upper_body_split_form_with_tendon_helpers_matches_oracle():
    """The library's form (round 4): the two arms hand their tendons to a helper wave each - kinematics of the spine and the arm
    from the stage state the parts publish, the arm's 13 tendons, wrench sums per link back through the exchange area, three
    barriers per acceleration - which takes the tendons off the longest parts' path."""
    from gym_roboy_amd.envs.robots import UpperBodyRobot
    import gen_tree_lane_baked as gen
    plain = check_split(UpperBodyRobot().get_description(), "upper_body")
    info = check_split(UpperBodyRobot().get_description(), "upper_body_h2", max_helpers=2)       # the generator's default share of the tendons
    assert info["n_parts"] == 3 and info["n_helpers"] == 2
    assert info["max_stmt"] < 0.95 * plain["max_stmt"] and info["helper_stmt"] < 0.2 * info["max_stmt"]
    # every tendon of the arms to the helpers (share 100 %): 3 106 -> 2 552 statements on the longest part, 885 per helper
    accel, all_ = host_split_accel(UpperBodyRobot().get_description(), "upper_body_h2_all", 4, 2 | (100 << 8))
    assert all_["max_stmt"] < 0.85 * plain["max_stmt"] and all_["helper_stmt"] > info["helper_stmt"]
    one = check_split(UpperBodyRobot().get_description(), "upper_body_h1", max_helpers=1)
    assert one["n_helpers"] == 1


def _sums_of_two_single_use_products(text):
    """Statements `tA +- tB` of the generated text whose operands are both products used nowhere else: there a compiler that may
    contract has a choice (which product to fuse), and made it differently in different kernels around the same text."""
    import collections
    import re
    stm = re.findall(r"const (?:float|rbl_f2) (t\d+) = ([^;]+);", text)
    defs = dict(stm)
    uses = collections.Counter(u for _, expr in stm for u in re.findall(r"\bt\d+\b", expr))
    product = lambda x: x in defs and "rbl_" not in defs[x] and re.fullmatch(r"[^+\-]*\*[^+\-]*", defs[x].replace("(-", "(")) is not None
    n = 0
    for _, expr in stm:
        m = re.fullmatch(r"\(?-?(t\d+)\)? [+-] \(?-?(t\d+)\)?", expr.strip())
        if m and product(m.group(1)) and product(m.group(2)) and uses[m.group(1)] == 1 and uses[m.group(2)] == 1:
            n += 1
    return n


def test_generated_text_leaves_no_contraction_to_choose():
    """a*b + c*d is written as rbl_fma(a, b, c*d) everywhere (dot products, cross products, the tendons' activation): the plain step
    and the fused env step of a robot - separate compilations of this text - then agree bit for bit (GPU:
    tests/test_random_robots_gpu.py, the hiprtc-built kernels of a random robot)."""
    import gen_tree_lane_baked as gen
    from gym_roboy_amd.envs.robots import UpperBodyRobot, RobotDescription
    from random_robots import random_tree_spec
    os.makedirs(BUILD, exist_ok=True)
    descs = [UpperBodyRobot().get_description()] + [RobotDescription(random_tree_spec(s)) for s in (4, 9)]
    for k, desc in enumerate(descs):
        one = os.path.join(BUILD, "contract_one_%d.hpp" % k)
        gen.generate(desc, one)
        text = open(one).read()
        assert "rbl_fma(" in text and _sums_of_two_single_use_products(text) == 0
        split = os.path.join(BUILD, "contract_split_%d.hpp" % k)
        gen.generate_split(desc, split, 4, 2, 80, 1, 0, 1)
        assert _sums_of_two_single_use_products(open(split).read()) == 0
    assert _sums_of_two_single_use_products("const float t1 = a * b;\nconst float t2 = c * d;\nconst float t3 = t1 - t2;") == 1    # (the scan sees one)


TWO_SWEEPS, CUT, SHARE_TRUNK = 1 << 16, 1 << 17, 1 << 18          # bits of the generator entry's max_helpers word (csrc/gen_tree_lane.cpp)


def test_upper_body_split_form_with_the_backward_pass_in_two_sweeps_matches_oracle():
    """The library's form: the bias-force recursion is linear in the forces, so the arms run the whole backward pass WITHOUT the
    tendon wrenches before barrier T (beside the helpers' tendon work) and propagate only the wrenches' part behind it."""
    from gym_roboy_amd.envs.robots import UpperBodyRobot
    import gen_tree_lane_baked as gen
    desc = UpperBodyRobot().get_description()
    lib_form = gen.SPLIT_HELPERS | (gen.SPLIT_HELPER_SHARE << 8) | (TWO_SWEEPS if gen.SPLIT_TWO_SWEEPS else 0) | (CUT if gen.SPLIT_CUT else 0)
    info = check_split(desc, "upper_body_lib_form", max_helpers=lib_form)
    one = check_split(desc, "upper_body_h2_s70", max_helpers=2 | (70 << 8))
    two = check_split(desc, "upper_body_h2_s70_t", max_helpers=2 | (70 << 8) | TWO_SWEEPS)
    assert info["n_parts"] == 3 and two["n_helpers"] == 2
    # what is left behind T: the text of the longest part between its second and third workgroup barrier
    def behind_t(tag):
        text = open(os.path.join(BUILD, "lane_split_%s.hpp" % tag)).read()
        body = text.split("RBL_FN void rbl_part0(")[1].split("\n}\n")[0].split("RBL_PART_BARRIER;")
        return sum(1 for line in body[2].split("\n") if " = " in line and not line.strip().startswith("//"))
    assert behind_t("upper_body_h2_s70_t") < 0.25 * behind_t("upper_body_h2_s70")


@pytest.mark.parametrize("seed", [4, 9, 10])
def test_random_robots_split_form_in_two_sweeps(seed):
    from gym_roboy_amd.envs.robots import RobotDescription
    from random_robots import random_tree_spec
    check_split(RobotDescription(random_tree_spec(seed)), "random%dt" % seed, n=6, tol=5e-4, max_helpers=2 | (100 << 8) | TWO_SWEEPS)


def test_one_part_can_evaluate_the_trunk_links_inertias_for_all():
    """share_trunk (measured, within the noise, not enabled in the library): the lightest part publishes the trunk links' inertias, bias
    forces and velocity products; the others run the trunk's frames only and fetch them behind barrier X - the same values in every
    part, so the trunk's accelerations still come out bit-identical (check_split counts mismatches)."""
    from gym_roboy_amd.envs.robots import UpperBodyRobot, RobotDescription
    from random_robots import random_tree_spec
    desc = UpperBodyRobot().get_description()
    base = check_split(desc, "upper_body_h2_s70_t", max_helpers=2 | (70 << 8) | TWO_SWEEPS)
    shared = check_split(desc, "upper_body_h2_s70_t_st", max_helpers=2 | (70 << 8) | TWO_SWEEPS | SHARE_TRUNK)
    assert shared["max_stmt"] < base["max_stmt"] and shared["x_slots"] > base["x_slots"]
    check_split(desc, "upper_body_st", max_helpers=SHARE_TRUNK)                       # ... and without helpers (one barrier per acceleration)
    for seed in (4, 9):
        check_split(RobotDescription(random_tree_spec(seed)), "random%dst" % seed, n=6, tol=5e-4, max_helpers=2 | (70 << 8) | TWO_SWEEPS | SHARE_TRUNK)


def test_upper_body_cut_form_matches_oracle():
    """The cut form (measured, not selected: csrc/roboy_sim.hip RB_SPLIT_CUT): each arm as a proximal wave (three links, the tendons) and
    a distal one (four links) that is a part of its own; five barriers per acceleration."""
    from gym_roboy_amd.envs.robots import UpperBodyRobot
    desc = UpperBodyRobot().get_description()
    info = check_split(desc, "upper_body_cut", max_helpers=2 | CUT)
    assert info["n_parts"] == 5 and info["n_helpers"] == 0
    assert info["part_of_joint"][6:13] == [0, 0, 0, 3, 3, 3, 3] and info["part_of_joint"][13:20] == [1, 1, 1, 4, 4, 4, 4]
    assert info["max_stmt"] < 0.7 * 2875                      # ... of the helper form's longest part (70 % share, two sweeps)
    text = open(os.path.join(BUILD, "lane_split_upper_body_cut.hpp")).read()
    assert "#define RBL_X_SINGLE 1" in text and "#define RBL_ACC_JOINTS 7" in text
    for part in range(5):                                     # the same number of workgroup barriers in every wave
        body = text.split("RBL_FN void rbl_part%d(" % part)[1].split("\n}\n")[0]
        assert body.count("RBL_PART_BARRIER;") == 5
    check_split(desc, "upper_body_cut_s40", max_helpers=2 | (40 << 8) | CUT)      # the distal waves take 40 % of the tendons


@pytest.mark.parametrize("seed", [4, 5, 9, 10])
def test_random_robots_cut_form(seed):
    from gym_roboy_amd.envs.robots import RobotDescription
    from random_robots import random_tree_spec
    desc = RobotDescription(random_tree_spec(seed))
    check_split(desc, "random%dc" % seed, n=6, tol=5e-4, max_helpers=2 | CUT)
    check_split(desc, "random%dcs" % seed, n=6, tol=5e-4, max_helpers=2 | (40 << 8) | CUT)


def test_upper_body_split_in_two_parts():
    from gym_roboy_amd.envs.robots import UpperBodyRobot
    assert check_split(UpperBodyRobot().get_description(), "upper_body2", max_parts=2)["n_parts"] == 2


@pytest.mark.parametrize("seed", [4, 5, 9, 10, 0])
def test_random_robots_split_form(seed):
    """Random trees: several roots (no trunk), branches tied together by tendons (merged into one part), trunks of several links."""
    from gym_roboy_amd.envs.robots import RobotDescription
    from random_robots import random_tree_spec
    import gen_tree_lane_baked as gen
    desc = RobotDescription(random_tree_spec(seed))
    try:
        check_split(desc, "random%d" % seed, n=6, tol=5e-4)
        check_split(desc, "random%dh" % seed, n=6, tol=5e-4, max_helpers=2)      # ... and with tendon helpers for the longest parts
    except RuntimeError as exc:                                # no split form for this robot: a chain, or everything tied into one group
        assert "rb_gen_tree_lane_split failed" in str(exc)
        pytest.skip("robot %d has no split form" % seed)


def test_star_robot_splits_into_the_maximum_number_of_parts():
    from gym_roboy_amd.envs.robots import RobotDescription
    from random_robots import random_tree_spec
    spec = random_tree_spec(77, n_q=9, n_t=8, shape="star")
    for t in spec["tendons"]:                                  # tendons between the base and ONE link each: nothing ties branches together
        link = t["via_points"][-1]["link"]
        link = link if link >= 1 else 1
        t["via_points"] = [{"link": -1, "pos": t["via_points"][0]["pos"]}, {"link": link, "pos": t["via_points"][-1]["pos"]}]
    info = check_split(RobotDescription(spec), "star9", n=6, tol=5e-4)
    assert info["n_parts"] == 4 and info["part_of_joint"][0] == -1
    info = check_split(RobotDescription(spec), "star9h", n=6, tol=5e-4, max_helpers=3)
    assert info["n_parts"] == 4 and 1 <= info["n_helpers"] <= 3


@pytest.mark.parametrize("seed", [0, 2])
def test_device_definitions_accept_every_pair_constant_shape_the_generator_writes(seed):
    """Robots with two structurally identical branches are written as ONE stream of pair values; a pair CONSTANT in that stream is a
    type of its own on the device (tree_lane_defs.hpp: rbl_k2c - packed or per half, by the accessor the function is instantiated
    with) and must combine with pair values and plain floats in every shape the generator emits (`K2 - t`, `t * K2`,
    `rbl_fma(t, K2, K2)`, ...).  The generated text of the mirrored random robots the GPU tests build at run time, compiled here for
    gfx950 by hiprtc in BOTH modes (-DRBL_K2_SPLIT=0 / 1; round 6: a shape without an overload failed on the GPU box only)."""
    import gen_tree_lane_baked as gen
    from gym_roboy_amd.envs.robots import RobotDescription
    from random_robots import random_mirrored_spec
    hdr = os.path.join(BUILD, "lane_rtc_mirrored_%d.hpp" % seed)
    gen.generate(RobotDescription(random_mirrored_spec(seed, n_branch=3 + seed, n_t_branch=3 + seed)), hdr)
    text = open(hdr).read()
    assert "RBL_K2(" in text and "rbl_f2" in text
    text = text[:text.rindex("#define RBL_TEXT_HASH")]
    for mode in (0, 1):
        src = '#define RBL_K2_SPLIT %d\n#include "tree_lane_defs.hpp"\n#define RBL_NS rbl_jit\n' % mode + text + '#include "tree_lane.hpp"\n'
        rc, log, size = _hiprtc_compile(src, "roboy_tree_lane_jit.hip", ["rbl_jit::tree_lane_step<0>"])
        assert rc == 0, log[:3000]
        assert size > 10000
