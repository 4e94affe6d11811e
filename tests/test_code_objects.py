"""What the shipped gfx950 code objects say (no GPU needed): libroboy_sim.so is unbundled and every kernel's metadata note is read
(tools/code_object_meta.py).  No kernel the library can pick by itself may use scratch memory or spill vector registers - a scratch
access is a full memory latency, and the joint-tree kernels run one wave per SIMD with nothing to cover it - outside an allow-list
with a reason per entry.  Also: the late-read argument of the ball-joint env kernels sits where the kernels look for it."""
import os
import re
import subprocess
import sys
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import code_object_meta as com  # noqa: E402

LIB = os.path.join(ROOT, "gym_roboy_amd", "csrc", "libroboy_sim.so")

# (regex on the short kernel name, what is allowed, why).  allowed: "agpr-spills" = vector-register spills that live in AGPRs (no
# private segment); "scratch" = a private segment that the code really uses; "phantom" = a private segment in the metadata that NO
# instruction of the kernel touches (checked by disassembly below).
ALLOW = [
    (r"^rbl_baked::tree_lane_(env_)?step<[01]>$", "agpr-spills",
     "one wave per SIMD owns the whole 512-entry file; the compiler parks a few of the ~470 live values in AGPRs (v_accvgpr moves, no memory)"),
    (r"^rbl_split2_baked::tree_split_(env_)?step<[01]>$", "agpr-spills",
     "the lean two-part split form: parking slots and RK4 sums in registers by design, 2-4 values end up in AGPRs (no memory)"),
    (r"^rbt::tree_(env_)?step_aba<[01], 2, (true|false)>$", "scratch",
     "the octet kernels: the FALLBACK form for joint trees without generated code, capped at 128 registers so that two workgroups share a "
     "SIMD; 8-120 bytes of scratch per lane, measured cost inside their 50-60 us per step (profiles/r2_a); AUTO prefers generated code"),
    (r"^rbk::msj_env_step_kernel<0, 64, 8, rb::MsjConst<float, 8>, false>$", "phantom",
     "kernarg constants, tendon loop written out: 2 scalar spills into a vector-register lane; the register allocator leaves a 132-byte "
     "private segment behind that no instruction touches (ROCm 7.2 / LLVM: spill slots allocated, then all of them served by lanes)"),
]


@pytest.fixture(scope="module")
def meta():
    if not os.path.exists(LIB):
        pytest.fail("libroboy_sim.so is not built (run __graft_entry__.build())")
    return {com.short(k): v for k, v in com.kernel_metadata(LIB).items()}


def _allowed(name):
    for pat, what, why in ALLOW:
        if re.match(pat, name):
            assert len(why) > 40
            return what
    return None


def test_every_kernel_of_the_library_is_listed_with_its_resources(meta):
    assert len(meta) >= 85
    for name, m in meta.items():
        for field in ("vgpr_count", "agpr_count", "sgpr_count", "private_segment_fixed_size", "group_segment_fixed_size", "vgpr_spill_count"):
            assert field in m, (name, field)
        assert m["vgpr_count"] <= 512 and m.get("wavefront_size", 64) == 64
    # the headline kernel: 64 registers class (8 waves per SIMD possible), no scratch, no spills of any kind
    head = meta["rbk::msj_step_env_per_lane_rs<1, 256, true>"]
    assert head["private_segment_fixed_size"] == 0 and head["vgpr_spill_count"] == 0 and head["sgpr_spill_count"] == 0 and head["vgpr_count"] <= 64


def test_no_scratch_and_no_vector_register_spills_outside_the_allow_list(meta):
    offenders = []
    for name, m in sorted(meta.items()):
        scratch, vspill = m["private_segment_fixed_size"], m["vgpr_spill_count"]
        if scratch == 0 and vspill == 0:
            continue
        what = _allowed(name)
        if what == "agpr-spills" and scratch == 0:
            continue
        if what in ("scratch", "phantom"):
            continue
        offenders.append("%s: %d bytes of scratch per lane, %d vector-register spills" % (name, scratch, vspill))
    assert not offenders, "\n".join(offenders)


def test_allow_list_entries_still_exist_and_still_need_their_entry(meta):
    """An entry whose kernels are clean again (or gone) must leave the list: it would hide the next regression."""
    for pat, what, why in ALLOW:
        hit = [n for n in meta if re.match(pat, n)]
        assert hit, "allow-list entry matches no kernel: %s" % pat
        dirty = [n for n in hit if meta[n]["private_segment_fixed_size"] or meta[n]["vgpr_spill_count"]]
        assert dirty, "allow-list entry no longer needed: %s" % pat


def _disassemble(image):
    with tempfile.NamedTemporaryFile(suffix=".co") as fh:
        fh.write(image)
        fh.flush()
        return subprocess.run([os.path.join(com.LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", fh.name], capture_output=True, text=True,
                              check=True).stdout


def test_phantom_private_segments_are_untouched_by_the_code(meta):
    """'phantom' entries: the kernel's instruction stream holds no scratch instruction and no access through the private-segment buffer."""
    phantoms = [n for n in meta if _allowed(n) == "phantom" and meta[n]["private_segment_fixed_size"]]
    if not phantoms:
        pytest.skip("no phantom private segment in this build")
    text = "\n".join(_disassemble(img) for img in com.code_objects(LIB))
    for name in phantoms:
        sym = meta[name]["symbol"]
        body = text.split("<%s>:" % sym, 1)[1]
        body = body.split("\n\n", 1)[0] if "\n\n" in body else body
        body = re.split(r"\n[0-9a-f]+ <", body, 1)[0]
        assert len(body.splitlines()) > 100
        assert not re.search(r"\bscratch_(load|store)", body), name
        assert not re.search(r"buffer_(load|store)_\w+ [^\n]*, s\[0:3\]", body), name      # (the private-segment resource of older ABIs)
        assert "s_endpgm" in body


def test_the_late_env_argument_sits_where_the_kernels_read_it():
    """msj_kernels.hpp: the ball-joint env kernels with kernarg constants read their MsjEnvArgs argument behind the step through the
    kernel-argument segment at msj_env_args_offset(bytes of the leading arguments).  The code objects list every argument's offset
    and size: the last explicit by-value argument must lie at that offset."""
    sizes = {"MsjEnvArgs": None}
    notes = []
    for image in com.code_objects(LIB):
        with tempfile.NamedTemporaryFile(suffix=".co") as fh:
            fh.write(image)
            fh.flush()
            notes.append(subprocess.run([os.path.join(com.LLVM, "llvm-readelf"), "--notes", fh.name], capture_output=True, text=True, check=True).stdout)
    text = "\n".join(notes)
    # per kernel: the explicit arguments (offset, size, kind) in order
    kernels = {}
    for block in text.split("\n  - .agpr_count:")[1:]:
        m = re.search(r"\.name:\s+(\S+)", block)
        if not m:
            continue
        args = []
        for entry in re.split(r"\n      - ", block.split(".args:", 1)[1].split("\n    .group_segment_fixed_size", 1)[0])[1:]:
            f = {k: v for k, v in re.findall(r"\.(offset|size|value_kind):\s+(\w+)", entry)}
            if not f["value_kind"].startswith("hidden"):
                args.append((int(f["offset"]), int(f["size"]), f["value_kind"]))
        kernels[m.group(1).strip("'\"")] = args
    checked = 0
    const8 = None
    for sym, args in kernels.items():
        if "msj_env_step" not in sym:
            continue
        last = args[-1]
        assert last[2] == "by_value" and last[1] % 8 == 0, (sym, last)
        sizes["MsjEnvArgs"] = sizes["MsjEnvArgs"] or last[1]
        assert last[1] == sizes["MsjEnvArgs"]                               # one struct, one size
        lead_end = max(a[0] + a[1] for a in args[:-1])
        assert last[0] == (lead_end + 7) // 8 * 8, (sym, args)               # = msj_env_args_offset(lead bytes), alignof(MsjEnvArgs) = 8
        if "tendon_per_lane" in sym:
            assert [a[2] for a in args] == ["by_value", "global_buffer", "by_value"]
        elif "mirror_pairs" in sym:
            assert [a[1] for a in args[:2]] == [args[0][1], 32]              # Const8, PairMap (8 ints)
        const8 = const8 or args[0][1]
        checked += 1
    assert checked >= 20 and const8 is not None
