import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "navigation-by-deja-vu_amd"))
import numpy as np
import navsim_amd
from oracle import oracle
from tests.test_gpu_properties import build
rng = np.random.default_rng(int.from_bytes(os.urandom(4), "little"))
eng = navsim_amd.FamiliarityEngine(0)
kinds = ["random", "levels", "two_hues", "few_hues", "coarse"]
n_res = n_ovf = 0
for it in range(3000):
    seed = int(rng.integers(0, 10**9)); F = int(rng.integers(1, 400)); h = int(rng.integers(1, 14)); w = int(rng.integers(1, 14))
    A = int(rng.integers(1, 65)); cw = [0.0, 0.25, 0.5, 1.0, float(rng.random())][int(rng.integers(0, 5))]
    kind = kinds[int(rng.integers(0, 5))]
    lib, pat = build(seed, F, h, w, A, kind)
    want = oracle.step(lib, pat, cw)
    eng.set_library(lib, cw)
    got = eng.step(pat, want_scene=True)
    ok = got["best_idex"] == want["best_idex"] and got["best_view"] == want["best_view"] and \
         np.allclose(got["angle_familiarity"], want["angle_familiarity"], rtol=1e-9, atol=1e-12) and \
         np.allclose(got["scene_familiarity"], want["scene_familiarity"], rtol=1e-9, atol=1e-12)
    n_res += bool(got["flags"] & 1); n_ovf += bool(got["flags"] & 4)
    if not ok:
        print("MISMATCH", seed, F, h, w, A, cw, kind, got["best_idex"], want["best_idex"], got["best_view"], want["best_view"], got["flags"], got["n_candidates"])
        break
else:
    print("3000 random problems ok; resolver ran in", n_res, "overflow fallback in", n_ovf)
