"""Random multi-block problems with duplicated views and repeated headings (ties across k_finish blocks, the shared
extra-candidate list, resolver and overflow paths): HIP decisions vs the oracle.
usage: python tests/manual/stress_blocks.py [cases] [seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "navigation-by-deja-vu_amd"))
import numpy as np
import navsim_amd
from navsim_amd import synth
from oracle import oracle


def make_case(rng):
    F = int(rng.integers(257, 3000)); h = int(rng.integers(2, 9)); w = int(rng.integers(2, 9))
    A = int(rng.integers(1, 65)); cw = [0.0, 0.25, 0.5, 1.0, float(rng.random())][int(rng.integers(0, 5))]
    seed = int(rng.integers(0, 10**9))
    lib = synth.synth_views(seed, F, h, w)
    pat = synth.synth_patches(seed, A, h, w)
    # duplicates of a few views scattered over the library (other blocks included), sometimes a long run of them
    for _ in range(int(rng.integers(0, 4))):
        src = int(rng.integers(0, F))
        n = int(rng.integers(1, 6)) if rng.random() < 0.8 else int(rng.integers(50, 600))
        idx = rng.integers(0, F, size=n)
        lib[idx] = lib[src]
        if rng.random() < 0.8:
            hs = rng.integers(0, A, size=int(rng.integers(1, 4)))
            pat[hs] = lib[src] if rng.random() < 0.7 else synth.near_match_patch(lib[src], seed + 1, 0.05)
    return lib, pat, cw


def run(cases, seed):
    rng = np.random.default_rng(seed)
    eng = navsim_amd.FamiliarityEngine(0)
    n_res = n_ovf = 0
    for it in range(cases):
        lib, pat, cw = make_case(rng)
        want = oracle.step(lib, pat, cw)
        eng.set_library(lib, cw)
        got = eng.step(pat, want_scene=True)
        ok = got["best_idex"] == want["best_idex"] and got["best_view"] == want["best_view"] and \
            np.allclose(got["angle_familiarity"], want["angle_familiarity"], rtol=1e-9, atol=1e-12) and \
            np.allclose(got["scene_familiarity"], want["scene_familiarity"], rtol=1e-9, atol=1e-12)
        n_res += bool(got["flags"] & 1); n_ovf += bool(got["flags"] & 4)
        if not ok:
            print("MISMATCH case", it, lib.shape, pat.shape, cw, got["best_idex"], want["best_idex"], got["best_view"],
                  want["best_view"], got["flags"], got["n_candidates"])
            return 1
    print("%d multi-block problems ok; resolver ran in %d, overflow fallback in %d" % (cases, n_res, n_ovf))
    return 0


if __name__ == "__main__":
    sys.exit(run(int(sys.argv[1]) if len(sys.argv) > 1 else 1000, int(sys.argv[2]) if len(sys.argv) > 2 else 12345))
