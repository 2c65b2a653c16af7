#!/usr/bin/env python3
"""Register / scratch / occupancy table of every kernel in csrc/dejavu_hip.hip (hipcc -Rpass-analysis), to catch a
change that costs a scoring kernel its occupancy or sends it to scratch.  usage: python tools/kernel_resources.py [filter]"""
import os, re, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "navigation-by-deja-vu_amd", "csrc")
out = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-c",
                      "-Rpass-analysis=kernel-resource-usage", "dejavu_hip.hip", "-o", "/dev/null"], cwd=src,
                     capture_output=True, text=True).stderr
flt = sys.argv[1] if len(sys.argv) > 1 else ""
cur = None
rows = []
for line in out.splitlines():
    m = re.search(r"remark: (.*?) \[-Rpass", line)
    if not m:
        continue
    t = m.group(1).strip()
    if t.startswith("Function Name:"):
        name = subprocess.run(["c++filt", t.split(": ", 1)[1]], capture_output=True, text=True).stdout.strip()
        cur = {"name": re.sub(r"\(.*", "", name).replace("void dv::", "").replace("dv::", "")}
        rows.append(cur)
    elif cur is not None and ":" in t:
        k, v = t.split(":", 1)
        cur[k.strip()] = v.strip()
print("%-44s %5s %5s %7s %5s %6s" % ("kernel", "VGPR", "SGPR", "scratch", "occ", "LDS"))
for r in rows:
    if flt in r["name"]:
        print("%-44s %5s %5s %7s %5s %6s" % (r["name"][:44], r.get("VGPRs"), r.get("TotalSGPRs"), r.get("ScratchSize [bytes/lane]"),
                                             r.get("Occupancy [waves/SIMD]"), r.get("LDS Size [bytes/block]")))
