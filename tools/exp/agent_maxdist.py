#!/usr/bin/env python3
"""The agent's rate with the experiment's max_distance_to_training_path = 450 (scripts/run_experiment.py:40): the metrics' answers are
collected a step late while the triangle inequality keeps the stop of :264 impossible (DEJAVU_AGENT_PIPELINE=0: one call per step and
the answer awaited at once, as before)."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "navigation-by-deja-vu_amd"))
import numpy as np
import navsim_amd
from navsim_amd import synth
L, n_views = 2000, 50000
land = synth.synth_landscape(20261004, L, 4)
path = synth.sin_training_path(0.5, 0.2 * L, 0.6 * L, arclen=0.6 * L * 1.4 / n_views)[:n_views]
for md in (450.0, np.inf):
    nsf = navsim_amd.NavBySceneFamiliarity(land, (64, 64), 0.5, n_test_angles=16, n_sensor_levels=5, max_distance_to_training_path=md,
                                           familiarity_model=navsim_amd.sads_familiarity(0.25))
    nsf.train_from_path(path)
    d = path[2] - path[1]
    nsf.angle = float(np.arctan2(d[1], d[0]) % (2 * np.pi))
    nsf.position = path[1] + np.array([1.0, -1.0])
    for _ in range(50):
        nsf.step_forward()
    t0 = time.perf_counter()
    for _ in range(1000):
        nsf.step_forward()
    dt = (time.perf_counter() - t0) / 1000
    print("max_distance_to_training_path=%s, track_scene_familiarity=True: %.1f us per step (%.0f steps/s)" % (md, dt * 1e6, 1 / dt))
    nsf._engine.close()
