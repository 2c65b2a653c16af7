#!/usr/bin/env python3
"""Timing of the batched-agent path (BASELINE.json configs[4]: 256 agents, 64x64 sensor, 100k views, one GPU's share).

usage: python tools/bench_batch.py [--agents 32] [--views 100000] [--headings 16] [--chem-weight 0.25]
Each agent's near-match patch is planted on a different stored view, so every decision is checked.
"""
import argparse, os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "navigation-by-deja-vu_amd"))
import numpy as np
import navsim_amd
from navsim_amd import synth

ap = argparse.ArgumentParser()
ap.add_argument("--agents", type=int, default=32)
ap.add_argument("--views", type=int, default=100000)
ap.add_argument("--headings", type=int, default=16)
ap.add_argument("--sensor", type=int, default=64)
ap.add_argument("--chem-weight", type=float, default=0.25)
ap.add_argument("--steps", type=int, default=10)
args = ap.parse_args()
n, F, A, h = args.agents, args.views, args.headings, args.sensor
seed = 20261004
eng = navsim_amd.FamiliarityEngine(0)
eng.generate_library(seed, F, h, h, chem_weight=args.chem_weight)
patches = synth.synth_patches(seed, n * A, h, h).reshape(n, A, h, h, 3)
want = []
for g in range(n):
    f, a = (g * 7919 + 13) % F, (g * 5) % A
    patches[g, a] = synth.near_match_patch(synth.synth_views(seed, 1, h, h, first_view=f)[0], seed + g)
    want.append((a, f))
for _ in range(2):
    res = eng.step_batch(patches)
for g in range(n):
    assert (res[g]["best_idex"], res[g]["best_view"]) == want[g], (g, res[g]["best_idex"], res[g]["best_view"], want[g])
t0 = time.perf_counter()
for _ in range(args.steps):
    eng.step_batch(patches)
dt = (time.perf_counter() - t0) / args.steps
print("batch: %d agents x %d headings, %d views %dx%d: %.2f ms per ensemble step (patches uploaded each step), "
      "%.3g view-comparisons/s, %.0f agent-steps/s" % (n, A, F, h, h, dt * 1e3, n * A * F / dt, n / dt))
