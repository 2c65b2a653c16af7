#!/usr/bin/env python3
"""Timing of the ssd_f32 metric on BASELINE configs[1]'s shape (64x64, 50k views, 16 headings, float32 = 819 MB)."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "navigation-by-deja-vu_amd"))
import numpy as np
import navsim_amd

F, h, w, A = 50000, 64, 64, 16
rng = np.random.default_rng(1)
lib = rng.random((F, h, w), dtype=np.float32)
patches = rng.random((A, h, w), dtype=np.float32)
patches[7] = lib[31337] + np.float32(0.01)
eng = navsim_amd.FamiliarityEngine(0)
eng.set_library_f32(lib)
for _ in range(10):
    r = eng.step_f32(patches)
assert r["best_idex"] == 7 and r["best_view"] == 31337
eng.profile_kernel(True)
n = 100
t0 = time.perf_counter()
for _ in range(n):
    eng.step_f32(patches)
dt = time.perf_counter() - t0
ms, k = eng.profile_read()
print("ssd_f32: step %.1f us (patches uploaded each step), kernel %.1f us, %.0f GB/s (%.1f%% of 8 TB/s), %.3g view-comparisons/s"
      % (dt / n * 1e6, ms / k * 1e3, F * h * w * 4 / (ms / k * 1e-3) / 1e9, F * h * w * 4 / (ms / k * 1e-3) / 8e12 * 100, F * A * n / dt))
