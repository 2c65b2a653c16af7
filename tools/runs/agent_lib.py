"""What the sensed library of the agent benchmark looks like to the bit-plane planner (GPU run helper)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "navigation-by-deja-vu_amd"))
import navsim_amd
from navsim_amd import synth
L = 2000
land = synth.synth_landscape(20261004, L, 4)
path = synth.sin_training_path(0.5, 0.2 * L, 0.6 * L, arclen=0.6 * L * 1.4 / 5000)[:5000]
nsf = navsim_amd.NavBySceneFamiliarity(land, (64, 64), 0.5, n_test_angles=16, n_sensor_levels=5,
                                       familiarity_model=navsim_amd.sads_familiarity(0.25), track_scene_familiarity=False)
nsf.train_from_path(path)
mem = np.asarray(nsf.familiar_scenes) if hasattr(nsf, "familiar_scenes") else None
eng = nsf.familiarity_model.engine if hasattr(nsf.familiarity_model, "engine") else None
print("engine", eng)
for name in dir(nsf.familiarity_model):
    if "eng" in name.lower(): print("attr", name)
e = getattr(nsf.familiarity_model, "_engine", None) or getattr(nsf, "_engine", None)
if e is not None:
    print(e.library_info())
if mem is not None:
    for ch in range(3):
        print("channel", ch, np.unique(mem[..., ch])[:40])
