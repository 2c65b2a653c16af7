import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "navigation-by-deja-vu_amd"))
import numpy as np
import navsim_amd
from navsim_amd import synth
L = 2000
land = synth.synth_landscape(20261004, L, 4)
n_views = 50000
path = synth.sin_training_path(0.5, 0.2 * L, 0.6 * L, arclen=0.6 * L * 1.4 / n_views)[:n_views]
nsf = navsim_amd.NavBySceneFamiliarity(land, (64, 64), 0.5, n_test_angles=16, n_sensor_levels=5,
                                       familiarity_model=navsim_amd.sads_familiarity(0.25), track_scene_familiarity=False)
nsf.train_from_path(path)
d = path[2] - path[1]
for rep in range(3):
    for fake in (True, False):
        nsf.position = path[1] + np.array([1.0, -1.0]); nsf.angle = float(np.arctan2(d[1], d[0]) % (2 * np.pi)); nsf.reset_error()
        for _ in range(10): nsf.step_forward(fake=fake)
        t0 = time.perf_counter()
        for _ in range(300): nsf.step_forward(fake=fake)
        dt = time.perf_counter() - t0
        print("rep", rep, "fake", fake, "steps/s %.0f" % (300 / dt), "us/step %.1f" % (dt / 300 * 1e6), flush=True)
# where does the time go with fake=False
import cProfile, pstats
nsf.position = path[1] + np.array([1.0, -1.0]); nsf.angle = float(np.arctan2(d[1], d[0]) % (2 * np.pi)); nsf.reset_error()
pr = cProfile.Profile(); pr.enable()
for _ in range(300): nsf.step_forward(fake=False)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
