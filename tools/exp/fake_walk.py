#!/usr/bin/env python3
"""Debug aid: the bench's agent walk with fake=True, pipelined against one call per step: where do they differ, which step is slow."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "navigation-by-deja-vu_amd"))
import numpy as np
import navsim_amd
from navsim_amd import synth
def walk(pipe, n=400, fake=True):
    L = 2000
    land = synth.synth_landscape(20261004, L, 4)
    n_views = 50000
    path = synth.sin_training_path(0.5, 0.2 * L, 0.6 * L, arclen=0.6 * L * 1.4 / n_views)[:n_views]
    nsf = navsim_amd.NavBySceneFamiliarity(land, (64, 64), 0.5, n_test_angles=16, n_sensor_levels=5, familiarity_model=navsim_amd.sads_familiarity(0.25),
                                           track_scene_familiarity=False)
    nsf.pipeline_steps = pipe
    nsf.train_from_path(path)
    d = path[2] - path[1]
    nsf.angle = float(np.arctan2(d[1], d[0]) % (2 * np.pi))
    nsf.position = path[1] + np.array([1.0, -1.0])
    nsf.reset_error()
    out = []
    for t in range(n):
        t0 = time.perf_counter()
        nsf.step_forward(fake=fake)
        out.append((nsf.last_best_idex, float(nsf.position[0]), float(nsf.position[1]), float(nsf.angle), (time.perf_counter() - t0) * 1e6))
    nsf.clear_training()
    return out
import gc
gc_log = []
def _cb(phase, info):
    if phase == "start": gc_log.append([time.perf_counter(), info["generation"], None])
    else: gc_log[-1][2] = (time.perf_counter() - gc_log[-1][0]) * 1e6
gc.callbacks.append(_cb)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 400
for fake in (True, False):
    gc_log.clear()
    a = walk(True, n=N, fake=fake)
    print("fake", fake, "collections during the piped walk (gen, us) over 200 us:", [(g, int(us)) for _, g, us in gc_log if us and us > 200][:10], "of", len(gc_log))
    b = walk(False, n=N, fake=fake)
    idx = np.array([x[0] for x in a])
    same = float(np.mean(idx[1:] == idx[:-1]))
    vals, counts = np.unique(idx, return_counts=True)
    print("best heading repeats the last one in %.1f %% of the steps; histogram:" % (100 * same), dict(zip(vals.tolist(), counts.tolist())))
    diff = [i for i, (x, y) in enumerate(zip(a, b)) if x[:4] != y[:4]]
    print("fake", fake, "first difference:", diff[:3], " slow steps piped:", [(i, int(x[4])) for i, x in enumerate(a) if x[4] > 500][:6],
          " plain:", [(i, int(x[4])) for i, x in enumerate(b) if x[4] > 500][:6])
