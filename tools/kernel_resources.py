#!/usr/bin/env python3
"""Per-kernel resources of a built library: registers, LDS, SCRATCH (private segment) and spill counts, read from the AMDGPU metadata notes
of the gfx950 code objects embedded in the .so (section .hip_fatbin: one clang offload bundle per translation unit).

    python tools/kernel_resources.py [fldr-vfi_amd/libfldr_hip.so] [--scratch-only]

Used by tests/test_host_cpu.py::test_product_kernels_use_no_scratch: a kernel that starts spilling does not fail any numerical test, it
gets slower (round 6: three ring kernels, +18 %, found only by a same-box comparison with the previous round's build)."""
import os
import re
import struct
import subprocess
import sys
import tempfile

MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"


def code_objects(path, arch="gfx950"):
    data = open(path, "rb").read()
    out, pos = [], 0
    while True:
        pos = data.find(MAGIC, pos)
        if pos < 0:
            break
        n = struct.unpack_from("<Q", data, pos + len(MAGIC))[0]
        p = pos + len(MAGIC) + 8
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", data, p)
            triple = data[p + 24:p + 24 + tl].decode()
            p += 24 + tl
            if arch in triple and size:
                out.append(data[pos + off:pos + off + size])
        pos += len(MAGIC)
    return out


def kernels(path):
    """-> list of dicts (name, vgpr, agpr, sgpr, lds, scratch, vgpr_spills, sgpr_spills) over every gfx950 kernel in the library."""
    res = []
    for co in code_objects(path):
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(co)
            f.flush()
            txt = subprocess.run([READELF, "--notes", f.name], capture_output=True, text=True, check=True).stdout
        cur = None
        for l in txt.splitlines():
            m = re.match(r"\s*-?\s*\.(\w+):\s*(.*)$", l)
            if not m:
                continue
            k, v = m.group(1), m.group(2).strip().strip("'\"")
            if k == "agpr_count":                       # first key of a kernel's map (keys are sorted)
                cur = {"agpr": int(v)}
                res.append(cur)
            elif cur is not None:
                if k == "name":
                    cur["name"] = v
                elif k in ("vgpr_count", "sgpr_count", "group_segment_fixed_size", "private_segment_fixed_size", "vgpr_spill_count", "sgpr_spill_count"):
                    cur[{"vgpr_count": "vgpr", "sgpr_count": "sgpr", "group_segment_fixed_size": "lds", "private_segment_fixed_size": "scratch",
                         "vgpr_spill_count": "vgpr_spills", "sgpr_spill_count": "sgpr_spills"}[k]] = int(v)
    return [r for r in res if "name" in r]


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    lib = args[0] if args else os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "fldr-vfi_amd", "libfldr_hip.so")
    ks = kernels(lib)
    only = "--scratch-only" in sys.argv
    print("%d kernels in %s" % (len(ks), lib))
    for r in sorted(ks, key=lambda r: (-r.get("scratch", 0), r["name"])):
        if only and not r.get("scratch"):
            continue
        print("%-110s vgpr %3d agpr %3d sgpr %3d lds %6d scratch %4d spills v %3d s %3d" % (r["name"][:110], r.get("vgpr", -1), r["agpr"], r.get("sgpr", -1), r.get("lds", -1),
                                                                                         r.get("scratch", -1), r.get("vgpr_spills", -1), r.get("sgpr_spills", -1)))
