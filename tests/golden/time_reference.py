#!/usr/bin/env python3
"""how representative the oracle (C restatement) is of the reference's own CPU path.

Build container only (needs /root/reference, Cython, gcc): builds the reference's navsim/util.pyx in a scratch directory under
/tmp exactly as make_golden.py does, then times its sads_familiarity()(lib) per heading next to oracle.sads_hsv on the same
seeded inputs, one thread, and writes the times and their ratio to tests/golden/reference_timing.json (data only).

    python3 tests/golden/time_reference.py
"""
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, REPO)
import make_golden as mg          # noqa: E402
from navsim_amd import synth      # noqa: E402
from oracle import oracle         # noqa: E402


def main():
    work = mg.build_reference("/root/reference")
    mg.import_reference(work)
    import navsim.util as ref_util
    rows = []
    for name, (F, h, w, A) in (("C0 32x32/500/8", (500, 32, 32, 8)), ("C1 shape 64x64, F=5000 of 50000, 16 headings", (5000, 64, 64, 16)),
                               ("C2 shape 128x128, F=2000 of 500000, 4 of 32 headings", (2000, 128, 128, 4))):
        lib = synth.synth_views(7, F, h, w)
        patches = synth.synth_patches(7, A, h, w)
        for cw in (0.0, 0.25):
            func = ref_util.sads_familiarity(cw)(lib)
            fam_r, fam_o = np.empty(F), np.empty(F)
            func(patches[0], fam_r)
            oracle.sads_hsv(lib, patches[0], cw, fam_o)
            assert fam_r.tobytes() == fam_o.tobytes()
            t0 = time.perf_counter()
            for a in range(A):
                func(patches[a], fam_r)
            t_ref = time.perf_counter() - t0
            t0 = time.perf_counter()
            for a in range(A):
                oracle.sads_hsv(lib, patches[a], cw, fam_o)
            t_or = time.perf_counter() - t0
            rows.append(dict(config=name, chem_weight=cw, views=F, headings=A, reference_s=t_ref, oracle_s=t_or,
                             reference_cmp_per_s=F * A / t_ref, oracle_cmp_per_s=F * A / t_or, oracle_over_reference_time=t_or / t_ref))
            print(rows[-1])
    model = "unknown"
    with open("/proc/cpuinfo") as f:
        for line in f:
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    out = dict(what="reference navsim/util.pyx:31-73 (Cython 3, gcc -O2, as built by make_golden.py) against oracle/sads_oracle.c "
                    "(gcc -O2 -ffp-contract=off), one thread each, bit-identical outputs checked", host_cpu=model, rows=rows)
    with open(os.path.join(HERE, "reference_timing.json"), "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
