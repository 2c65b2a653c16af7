"""Random problems with more k_finish blocks than its last block folds in one round (70k-120k tiny views, <= 32
headings), duplicates scattered over the library: decisions vs the oracle.  usage: DEJAVU_FINISH=2 python tests/manual/stress_many_blocks.py [cases]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "navigation-by-deja-vu_amd"))
import numpy as np
import navsim_amd
from navsim_amd import synth
from oracle import oracle

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(5)
eng = navsim_amd.FamiliarityEngine(0)
n_res = 0
for it in range(cases):
    F = int(rng.integers(66000, 120000)); h = int(rng.integers(2, 5)); w = int(rng.integers(2, 5))
    A = int(rng.integers(1, 33)); cw = [0.0, 0.25, 0.5, 1.0][int(rng.integers(0, 4))]
    seed = int(rng.integers(0, 10**9))
    lib = synth.random_hsv(seed, (F, h, w, 3))
    pat = synth.random_hsv(seed + 1, (A, h, w, 3))
    for _ in range(int(rng.integers(0, 4))):
        src = int(rng.integers(0, F))
        lib[rng.integers(0, F, size=int(rng.integers(1, 40)))] = lib[src]
        pat[rng.integers(0, A, size=int(rng.integers(1, 3)))] = lib[src]
    want = oracle.step(lib, pat, cw)
    eng.set_library(lib, cw)
    got = eng.step(pat, want_scene=False)
    n_res += bool(got["flags"] & 1)
    if (got["best_idex"], got["best_view"]) != (want["best_idex"], want["best_view"]) or \
            not np.allclose(got["angle_familiarity"], want["angle_familiarity"], rtol=1e-9, atol=1e-12):
        print("MISMATCH", it, F, h, w, A, cw, got["best_idex"], want["best_idex"], got["best_view"], want["best_view"], got["flags"])
        sys.exit(1)
    if (it + 1) % 20 == 0:
        print("  ...%d ok" % (it + 1), flush=True)
print("%d many-block problems ok; resolver ran in %d" % (cases, n_res))
