#!/usr/bin/env python3
"""Summarise the rocprofv3 output of tools/profile_bench.sh into one JSON (kernel durations + HBM traffic).

HBM traffic per launch follows /opt/skills/guides/MI355X_MICROARCH.md (HBM section): FETCH_SIZE and WRITE_SIZE are
in KiB; on gfx950 FETCH_SIZE reports exactly half of the bytes of a wide coalesced streaming read (16 B/lane), so
the read side is doubled; WRITE_SIZE is exact for 4..16-byte-per-lane streaming stores.
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def biggest(pattern):
    """rocprofv3 writes one file set per process; the bench process has the largest."""
    f = glob.glob(pattern)
    return max(f, key=os.path.getsize) if f else None


def kernel_stats(d):
    f = biggest(d + "/trace/*/*kernel_stats.csv")
    out = []
    if f:
        for r in csv.DictReader(open(f)):
            out.append(dict(name=r["Name"].split("(")[0], calls=int(r["Calls"]), avg_us=float(r["AverageNs"]) / 1e3,
                            min_us=float(r["MinNs"]) / 1e3, max_us=float(r["MaxNs"]) / 1e3, pct=float(r["Percentage"])))
    return out


def counters(d, sub, kernel_substr):
    """Per-launch averages of the counters of ONE kernel: of the instantiations whose name contains `kernel_substr`
    the one dispatched most often (the timed steps' workgroup shape, not the few launches that time the others)."""
    f = biggest(d + "/" + sub + "/*/*counter_collection.csv")
    per_name = defaultdict(lambda: defaultdict(list))
    if f:
        for r in csv.DictReader(open(f)):
            if kernel_substr in r["Kernel_Name"]:
                per_name[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    if not per_name:
        return {}
    name = max(per_name, key=lambda n: max(len(v) for v in per_name[n].values()))
    out = {k: sum(v) / len(v) for k, v in per_name[name].items()}
    out["kernel_name"] = name.split("(")[0]
    return out


def main():
    d = sys.argv[1]
    ks = kernel_stats(d)
    out = dict(kernels=ks)
    for kern in ("k_sad_mfma", "k_sad_lc22", "k_patch_prep", "k_fold", "k_sad_tiles", "k_sad_packed", "k_sad_generic", "k_ssd_tiles", "k_ssd_f32_mfma", "k_ssd_f32_bf16x2", "k_ssd_u8_mfma", "k_finish", "k_combine",
                 "k_tail", "k_resolve_f32", "k_cand_f32x"):
        c = {}
        for sub in ("pmc_fetch", "pmc_write", "pmc_sq", "pmc_inst", "pmc_sqc"):
            c.update(counters(d, sub, kern))
        if "FETCH_SIZE" in c:
            c["hbm_read_bytes_per_launch"] = c["FETCH_SIZE"] * 1024 * 2      # gfx950: FETCH_SIZE counts half
        if "WRITE_SIZE" in c:
            c["hbm_write_bytes_per_launch"] = c["WRITE_SIZE"] * 1024
        if "hbm_read_bytes_per_launch" in c and "hbm_write_bytes_per_launch" in c:
            c["hbm_traffic_bytes_per_launch"] = c["hbm_read_bytes_per_launch"] + c["hbm_write_bytes_per_launch"]
        if c:
            out[kern] = c
    try:
        out["bench_under_trace"] = json.loads(open(d + "/bench_under_trace.json").read().strip().splitlines()[-1])
    except Exception as e:   # noqa: BLE001
        out["bench_under_trace"] = str(e)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
