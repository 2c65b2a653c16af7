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
eng = nsf._engine
for rep in range(4):
    for fake in (True, False):
        nsf.position = path[1] + np.array([1.0, -1.0]); nsf.angle = float(np.arctan2(d[1], d[0]) % (2 * np.pi)); nsf.reset_error()
        for _ in range(10): nsf.step_forward(fake=fake)
        t_enq = t_wait = 0.0
        t0 = time.perf_counter()
        for _ in range(300): nsf.step_forward(fake=fake)
        dt = time.perf_counter() - t0
        print("rep", rep, "fake", fake, "us/step %.1f" % (dt / 300 * 1e6), flush=True)
# split of the metrics calls
nsf.reset_error()
ts = {"enq": 0.0, "wait": 0.0}
for i in range(200):
    nsf.step_forward(fake=True)
    t0 = time.perf_counter(); eng.path_error_enqueue(nsf.position[0], nsf.position[1], 0.4); ts["enq"] += time.perf_counter() - t0
    nsf.step_forward(fake=True)
    t0 = time.perf_counter(); eng.path_error_wait(); ts["wait"] += time.perf_counter() - t0
print({k: "%.1f us" % (v / 200 * 1e6) for k, v in ts.items()})
