#!/usr/bin/env python3
"""Known-answer stress of the step's reductions across many workgroups (k_tail's arrival ticket and the per-heading
first-view atomics): 50 000 views = 196 k_tail blocks, a stored view planted under a random heading every step, the
decision must name exactly that heading and view.  usage: python tools/stress_tail.py [steps]"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "navigation-by-deja-vu_amd"))
import numpy as np
import navsim_amd
from navsim_amd import synth

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
F, h, A, seed = 50000, 64, 16, 20261004
eng = navsim_amd.FamiliarityEngine(0)
eng.generate_library(seed, F, h, h, chem_weight=0.25)
rng = np.random.default_rng(1)
base = synth.synth_patches(seed, A, h, h)
bad = 0
for it in range(steps):
    a, f = int(rng.integers(0, A)), int(rng.integers(0, F))
    pats = base.copy()
    pats[a] = synth.synth_views(seed, 1, h, h, first_view=f)[0]
    if it % 3 == 0:                       # a second heading sees the same view: the first heading must win
        a2 = int(rng.integers(0, A))
        pats[a2] = pats[a]
        a = min(a, a2)
    r = eng.step(pats, want_scene=False)
    if (r["best_idex"], r["best_view"]) != (a, f) or r["step_familiarity"] != float(h * h) or r["angle_view"][a] != f:
        bad += 1
        print("MISMATCH at step", it, "want", (a, f), "got", (r["best_idex"], r["best_view"]), r["step_familiarity"], r["flags"])
        if bad > 5:
            break
    if (it + 1) % 100000 == 0:
        print("  ...%d steps, %d mismatches" % (it + 1, bad), flush=True)
print("%d steps, %d mismatches" % (it + 1, bad))
sys.exit(1 if bad else 0)
