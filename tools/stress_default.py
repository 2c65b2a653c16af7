#!/usr/bin/env python3
"""Randomised stress of the step as it ships (matrix cores: fp4 / int8 chosen on the device, fused finishing, k_fold;
byte path where the tuner prefers it) against the CPU oracle: random library sizes and sensor shapes, headings,
chem_weight, planted duplicates of the best view (ties the exact resolver must settle), patches on and off the library's
levels, per-heading maxima and first views.  Not a pytest (minutes of GPU time): python tools/stress_default.py [problems]"""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "navigation-by-deja-vu_amd"))
sys.path.insert(0, REPO)
import numpy as np
import navsim_amd
from navsim_amd import synth
from oracle import oracle

n_problems = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(20261004)
bad = 0
t0 = time.time()
for it in range(n_problems):
    F = int(rng.choice([int(rng.integers(1, 400)), int(rng.integers(400, 9000)), int(rng.integers(9000, 70000)), int(rng.integers(70000, 160000))]))
    h, w = int(rng.integers(2, 20)), int(rng.integers(2, 20))
    if F > 20000:
        h, w = min(h, 8), min(w, 8)                    # keep the oracle in seconds
    A = int(rng.choice([1, 2, 7, 8, 16, 17, 32, 33, 64]))
    cw = float(rng.choice([0.0, 0.25, 0.5, 1.0, float(rng.random())]))
    lib = synth.synth_views(1000 + it, F, h, w)
    pats = synth.synth_patches(2000 + it, A, h, w)
    kind = it % 4
    if kind == 1:                                      # one byte off its level: the int8 form in the same launch
        pats[int(rng.integers(0, A)), int(rng.integers(0, h)), int(rng.integers(0, w)), 2] ^= 0x08
    if kind >= 2 and F > 3:                            # a unique best view, duplicated, seen under several headings
        star = lib[int(rng.integers(0, F))].copy()
        star[0, 0] = (77, 200, 13)
        for f in rng.choice(F, size=min(F, int(rng.integers(1, 5))), replace=False):
            lib[int(f)] = star
        for a in rng.choice(A, size=min(A, int(rng.integers(1, 4))), replace=False):
            pats[int(a)] = star
    want = oracle.step(lib, pats, cw)
    eng = navsim_amd.FamiliarityEngine(0)
    try:
        eng.set_library(lib, cw)
        for rep, want_scene in enumerate((False, True, False)):
            got = eng.step(pats, want_scene=want_scene)
            ok = (got["best_idex"], got["best_view"]) == (want["best_idex"], want["best_view"])
            ok = ok and np.allclose(got["angle_familiarity"], want["angle_familiarity"], rtol=1e-9, atol=1e-12)
            if want_scene:
                ok = ok and np.allclose(got["scene_familiarity"], want["scene_familiarity"], rtol=1e-9, atol=1e-12)
            if not ok:
                bad += 1
                print("MISMATCH problem", it, dict(F=F, h=h, w=w, A=A, cw=cw, kind=kind, rep=rep), "want", (want["best_idex"], want["best_view"]),
                      "got", (got["best_idex"], got["best_view"]), got["flags"], got["n_candidates"], eng.scoring_form(), flush=True)
                break
    finally:
        eng.close()
    if (it + 1) % 10 == 0:
        print("  %d problems, %d mismatches, %.0f s" % (it + 1, bad, time.time() - t0), flush=True)
print("%d problems, %d mismatches" % (n_problems, bad))
sys.exit(1 if bad else 0)
