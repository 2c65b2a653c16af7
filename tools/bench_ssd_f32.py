#!/usr/bin/env python3
"""Timing of the ssd_f32 metric: BASELINE configs[1]'s shape (64x64, 50k float32 views, 16 headings = 819 MB) and, with
`big`, configs[2] as BASELINE.json words it (128x128, 500k float32 views, 32 headings = 32.8 GB, generated on the device).
DEJAVU_SSD_MFMA=0 times the direct form (k_ssd_tiles) instead of the matrix-core form (k_ssd_f32_mfma).

    python tools/bench_ssd_f32.py [big] [steps]
"""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "navigation-by-deja-vu_amd"))
import numpy as np
import navsim_amd
from navsim_amd import synth

big = "big" in sys.argv[1:]
nums = [int(a) for a in sys.argv[1:] if a.isdigit()]
n = nums[0] if nums else (30 if big else 100)
F, h, w, A = (500000, 128, 128, 32) if big else (50000, 64, 64, 16)
rng = np.random.default_rng(1)
eng = navsim_amd.FamiliarityEngine(0)
eng.generate_library_f32(4242, F, h, w)
patches = rng.random((A, h, w), dtype=np.float32)
plant = 31337 % F
patches[7] = synth.synth_views_f32(4242, 1, h, w, first_view=plant)[0] + np.float32(0.01)
for _ in range(5):
    r = eng.step_f32(patches)
assert r["best_idex"] == 7 and r["best_view"] == plant, (r["best_idex"], r["best_view"])
eng.profile_kernel(True)
t0 = time.perf_counter()
for _ in range(n):
    eng.step_f32(patches)
dt = time.perf_counter() - t0
ms, k = eng.profile_read()
form = "direct (k_ssd_tiles)" if os.environ.get("DEJAVU_SSD_MFMA") == "0" else "matrix cores (k_ssd_f32_mfma)"
per = ms / k
byt = F * h * w * 4
print("ssd_f32 %dx%d x %d views x %d headings, %s: step %.1f us (patches uploaded each step), scoring %.1f us per step, "
      "%.0f GB/s on the library's %.3g bytes (%.1f%% of 8 TB/s), %.3g view-comparisons/s, candidates %d"
      % (w, h, F, A, form, dt / n * 1e6, per * 1e3, byt / (per * 1e-3) / 1e9, byt, byt / (per * 1e-3) / 8e12 * 100, F * A * n / dt,
         r["n_candidates"]))
eng.close()
