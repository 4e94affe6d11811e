#!/usr/bin/env python3
"""What the shipped code objects say about every kernel: registers, scratch, spills, static LDS - read from the AMDGPU metadata
notes of the gfx950 code objects embedded in a shared library (no GPU needed).

    python tools/code_object_meta.py [gym_roboy_amd/csrc/libroboy_sim.so] [--csv out.csv]

The library's `.hip_fatbin` section is a sequence of clang offload bundles (one per translation unit); each bundle's gfx950 entry is an
ELF whose NT_AMDGPU_METADATA note (`llvm-readelf --notes`) lists, per kernel: `.vgpr_count`, `.agpr_count`, `.sgpr_count`,
`.private_segment_fixed_size` (scratch bytes per lane), `.group_segment_fixed_size` (static LDS; dynamic LDS is a launch argument),
`.vgpr_spill_count`, `.sgpr_spill_count`, `.max_flat_workgroup_size`.  Used by tests/test_code_objects.py (no scratch in a kernel the
library can pick by itself) and by tools/summarize_profile.py (the resource columns of kernel_stats_by_grid.csv: rocprofv3's own
columns are the dispatch packet's granulated values, not the code object's)."""
import os
import re
import struct
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
FIELDS = ("vgpr_count", "agpr_count", "sgpr_count", "private_segment_fixed_size", "group_segment_fixed_size", "vgpr_spill_count",
          "sgpr_spill_count", "max_flat_workgroup_size", "kernarg_segment_size", "wavefront_size")


def code_objects(lib_path, arch="gfx950"):
    """The raw ELF images of every `arch` code object bundled into the library."""
    with tempfile.TemporaryDirectory() as tmp:
        fat = os.path.join(tmp, "fat.bin")
        subprocess.check_call([os.path.join(LLVM, "llvm-objcopy"), "--dump-section", ".hip_fatbin=" + fat, lib_path])
        with open(fat, "rb") as fh:
            data = fh.read()
    out = []
    for m in re.finditer(re.escape(MAGIC), data):
        p = m.start()
        (n,) = struct.unpack_from("<Q", data, p + len(MAGIC))
        q = p + len(MAGIC) + 8
        for _ in range(n):
            off, size, tlen = struct.unpack_from("<QQQ", data, q)
            q += 24
            triple = data[q:q + tlen].decode()
            q += tlen
            if arch in triple and size:
                out.append(data[p + off:p + off + size])
    return out


def demangle(names):
    if not names:
        return {}
    res = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True, check=True)
    return dict(zip(names, res.stdout.splitlines()))


def kernel_metadata(lib_path=None):
    """{demangled kernel name: {field: int, 'symbol': mangled name}} over every gfx950 code object of the library."""
    lib_path = lib_path or os.path.join(ROOT, "gym_roboy_amd", "csrc", "libroboy_sim.so")
    kernels = {}
    for image in code_objects(lib_path):
        with tempfile.NamedTemporaryFile(suffix=".co") as fh:
            fh.write(image)
            fh.flush()
            text = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", fh.name], capture_output=True, text=True, check=True).stdout
        # the YAML of amdhsa.kernels: entries start at "  - " (two spaces); keys are ".name:  value" lines indented by four
        cur = None
        in_kernels = False
        for line in text.splitlines():
            if line.startswith("amdhsa.kernels:"):
                in_kernels = True
                continue
            if in_kernels and re.match(r"^amdhsa\.\w+:", line):
                in_kernels = False
            if not in_kernels:
                continue
            m = re.match(r"^  (- | {2})\.(\w+):\s*(.*)$", line)
            if not m:
                continue
            if m.group(1) == "- ":
                cur = {}
                kernels_entry = cur
                kernels.setdefault("__list__", []).append(kernels_entry)
            if cur is None:
                continue
            key, val = m.group(2), m.group(3).strip()
            if key in FIELDS:
                cur[key] = int(val)
            elif key == "name":
                cur["symbol"] = val.strip("'\"")
    entries = kernels.pop("__list__", [])
    names = demangle([e["symbol"] for e in entries if "symbol" in e])
    return {names[e["symbol"]]: e for e in entries if "symbol" in e}


def short(name):
    """`void rbk::msj_step_env_per_lane_rs<1, 256, true>(rb::MsjConst<...>, ...)` -> `rbk::msj_step_env_per_lane_rs<1, 256, true>`"""
    name = re.sub(r"^void ", "", name).replace("(anonymous namespace)::", "")
    depth = 0
    for i, ch in enumerate(name):
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            return name[:i]
    return name


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    lib = args[0] if args else None
    meta = kernel_metadata(lib)
    rows = sorted(((short(k), v) for k, v in meta.items()), key=lambda kv: kv[0])
    if "--csv" in sys.argv:
        import csv
        with open(sys.argv[sys.argv.index("--csv") + 1], "w") as fh:
            w = csv.writer(fh)
            w.writerow(["Kernel", "VGPR", "AGPR", "SGPR", "ScratchBytesPerLane", "StaticLDSBytes", "VGPRSpills", "SGPRSpills", "MaxWorkgroup"])
            for k, v in rows:
                w.writerow([k, v.get("vgpr_count"), v.get("agpr_count"), v.get("sgpr_count"), v.get("private_segment_fixed_size"),
                            v.get("group_segment_fixed_size"), v.get("vgpr_spill_count"), v.get("sgpr_spill_count"), v.get("max_flat_workgroup_size")])
    print("%-92s %5s %5s %5s %8s %8s %6s %6s" % ("kernel", "vgpr", "agpr", "sgpr", "scratch", "lds", "vspill", "sspill"))
    for k, v in rows:
        print("%-92s %5d %5d %5d %8d %8d %6d %6d" % (k[:92], v.get("vgpr_count", -1), v.get("agpr_count", -1), v.get("sgpr_count", -1),
                                                      v.get("private_segment_fixed_size", -1), v.get("group_segment_fixed_size", -1),
                                                      v.get("vgpr_spill_count", -1), v.get("sgpr_spill_count", -1)))


if __name__ == "__main__":
    main()
