#!/usr/bin/env python3
"""Soak of k_sad_lc22 (passes of 64 headings: two view groups x two heading tiles per consumer, hand-counted waits, shared
accumulators) against the one-group body: two engines on the same library, fresh patches every iteration (on the library's levels,
off them, near-duplicates of stored views), every agent's record compared.  usage: stress_lc22.py [seconds]"""
import os, sys, time
REPO = os.path.dirname(os.path.abspath(__file__)); REPO = os.path.dirname(REPO)
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "navigation-by-deja-vu_amd"))
import numpy as np
import navsim_amd
from navsim_amd import synth
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
def engine(knob):
    old = os.environ.get("DEJAVU_LC22"); os.environ["DEJAVU_LC22"] = knob
    try: return navsim_amd.FamiliarityEngine(0)
    finally:
        if old is None: os.environ.pop("DEJAVU_LC22", None)
        else: os.environ["DEJAVU_LC22"] = old
t_end = time.time() + budget
rng = np.random.default_rng(20261005)
steps = agents = bad = 0
round_ = 0
while time.time() < t_end:
    round_ += 1
    h = w = int(rng.choice([8, 16, 24, 32]))
    F = int(rng.integers(41000, 90000))
    cw = float(rng.choice([0.0, 0.25, 0.5]))
    A = int(rng.choice([8, 16, 20, 32, 64]))
    n_agents = int(rng.integers(1, 1 + max(1, 192 // A)))
    lib = synth.synth_views(1000 + round_, F, h, w)
    e1, e0 = engine("1"), engine("0")
    try:
        for e in (e1, e0): e.set_library(lib, cw)
        t_round = time.time() + min(12.0, max(2.0, t_end - time.time()))
        it = 0
        while time.time() < t_round:
            it += 1
            p = synth.synth_patches(5000 + 97 * round_ + it, n_agents * A, h, w).reshape(n_agents, A, h, w, 3)
            kind = it % 3
            if kind == 1: p[..., 2] = synth.random_hsv(7000 + it, p.shape[:-1])             # off the levels: the int8 body
            if kind == 2:
                for k in range(min(n_agents, 3)): p[k, int(rng.integers(0, A))] = lib[int(rng.integers(0, F))]      # exact matches (ties with their duplicates, if any)
            r1, r0 = e1.step_batch(p), e0.step_batch(p)
            steps += 1; agents += n_agents
            ok = (r1.best_idex.tolist() == r0.best_idex.tolist() and r1.best_view.tolist() == r0.best_view.tolist() and
                  np.array_equal(r1.angle_familiarity, r0.angle_familiarity) and np.array_equal(r1.angle_view, r0.angle_view))
            if not ok:
                bad += 1
                print("MISMATCH round %d it %d: F=%d %dx%d cw=%g A=%d agents=%d kind=%d" % (round_, it, F, h, w, cw, A, n_agents, kind), flush=True)
    finally:
        e1.close(); e0.close()
    print("round %d: F=%d %dx%d cw=%g A=%d x %d agents, %d ensemble steps so far, %d mismatches" % (round_, F, h, w, cw, A, n_agents, steps, bad), flush=True)
print("stress_lc22: %d ensemble steps, %d agent records compared, %d mismatches" % (steps, agents, bad))
sys.exit(1 if bad else 0)
