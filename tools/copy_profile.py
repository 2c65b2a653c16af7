#!/usr/bin/env python3
"""Copies the judged evidence of a tools/profile_bench.sh run from gpurun_out/<tag>/ into profiles/.

usage: python tools/copy_profile.py r01
rocprofv3 writes one file set per process (bench.py starts helpers); the bench process is the largest.
"""
import glob, os, shutil, sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", tag)
dst = os.path.join(root, "profiles")
os.makedirs(dst, exist_ok=True)


def biggest(pattern):
    files = glob.glob(os.path.join(src, pattern))
    if not files:
        raise SystemExit("missing " + pattern)
    return max(files, key=os.path.getsize)


shutil.copy(os.path.join(src, "summary.json"), os.path.join(dst, tag + "_summary.json"))
shutil.copy(biggest("trace/*/*_kernel_stats.csv"), os.path.join(dst, tag + "_kernel_stats.csv"))
for name, out in (("pmc_fetch", "pmc_fetch_size"), ("pmc_write", "pmc_write_size")):
    # keep the scoring-path rows only: the full per-dispatch table is several MB
    keep = ("k_sad", "k_finish", "k_fold", "k_combine", "k_tail", "k_ssd", "k_exact")
    with open(biggest(name + "/*/*_counter_collection.csv")) as f, open(os.path.join(dst, f"{tag}_{out}.csv"), "w") as o:
        for i, line in enumerate(f):
            if i == 0 or any(k in line for k in keep):
                o.write(line)
print("copied", tag, "->", dst)
