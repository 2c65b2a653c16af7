#!/usr/bin/env python3
"""Register / scratch table of every kernel in the BUILT libdejavu_hip.so (what ships), read from the code object's
metadata -- to catch a change that sends a scoring kernel to scratch or costs it its occupancy.

    python tools/kernel_resources.py [filter]

`kernel_table()` is what tests/test_host_logic.py:test_shipped_scoring_kernels_use_no_scratch checks (no compile: the
gfx950 code object is unbundled from the library's .hip_fatbin section and its notes are parsed)."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "navigation-by-deja-vu_amd", "csrc", "libdejavu_hip.so")
LLVM = "/opt/rocm/lib/llvm/bin"


def kernel_table(lib=LIB):
    """[{name, vgpr, sgpr, scratch (bytes per lane), vgpr_spills, lds}] of the gfx950 kernels in `lib`."""
    with tempfile.TemporaryDirectory() as tmp:
        fat, co = os.path.join(tmp, "fat.bin"), os.path.join(tmp, "dev.co")
        subprocess.run([os.path.join(LLVM, "llvm-objcopy"), "--dump-section", ".hip_fatbin=" + fat, lib], check=True)
        subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o",
                        "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--input=" + fat, "--output=" + co], check=True)
        notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], check=True, capture_output=True,
                               text=True).stdout
    # amdhsa.kernels: one entry per kernel, opened by "  - .agpr_count:"; the fields wanted sit at the entry's own indent
    rows, cur = [], None
    keys = {".private_segment_fixed_size": "scratch", ".vgpr_count": "vgpr", ".sgpr_count": "sgpr",
            ".vgpr_spill_count": "vgpr_spills", ".group_segment_fixed_size": "lds", ".name": "mangled"}
    for line in notes.splitlines():
        if re.match(r"\s{2}- \.agpr_count:", line):
            cur = {}
            rows.append(cur)
            continue
        m = re.match(r"\s{4}(\.[a-z_]+):\s*(\S+)\s*$", line)
        if m and cur is not None and m.group(1) in keys:
            k, v = keys[m.group(1)], m.group(2)
            cur[k] = v if k == "mangled" else int(v)
    rows = [r for r in rows if "mangled" in r and "scratch" in r]
    names = subprocess.run(["c++filt"] + [r["mangled"] for r in rows], capture_output=True, text=True).stdout.splitlines()
    for r, n in zip(rows, names):
        r["name"] = re.sub(r"\(.*", "", n).replace("void dv::", "").replace("dv::", "")
    return rows


if __name__ == "__main__":
    flt = sys.argv[1] if len(sys.argv) > 1 else ""
    print("%-60s %5s %5s %7s %6s" % ("kernel", "VGPR", "SGPR", "scratch", "LDS"))
    for r in kernel_table():
        if flt in r["name"]:
            print("%-60s %5s %5s %7s %6s" % (r["name"][:60], r.get("vgpr"), r.get("sgpr"), r.get("scratch"), r.get("lds")))
