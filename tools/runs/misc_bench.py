"""cw = 0 (the reference's default) and 3-/4-hue libraries: kernel forms by timing, kernel time, bytes streamed."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "navigation-by-deja-vu_amd"))
import numpy as np
import navsim_amd
from navsim_amd import synth

def run(name, eng, A, steps=200):
    info = eng.library_info()
    for _ in range(20):
        eng.step_enqueue(want_scene=False); eng.step_wait(want_scene=False)
    eng.profile_kernel(True, every=4)
    t0 = time.perf_counter()
    for _ in range(steps):
        eng.step_enqueue(want_scene=False); eng.step_wait(want_scene=False)
    dt = (time.perf_counter() - t0) / steps
    ms, n = eng.profile_read(); eng.profile_kernel(False)
    shape = eng.workgroup_shape(A)
    streamed = info["bit_tile_bytes"] if shape == 6 else info["tile_bytes"]
    print("%-34s shape %d planes %d bits %d+%d  kernel %.1f us  step %.1f us  streamed %.1f MB -> %.2f TB/s" % (
        name, shape, info["n_planes"], info["bit_planes_hs"], info["bit_planes_v"], ms / n * 1e3, dt * 1e6, streamed / 1e6,
        streamed / (ms / n * 1e-3) / 1e12), flush=True)

os.environ["DEJAVU_VERBOSE"] = "1"
eng = navsim_amd.FamiliarityEngine(0)
for (F, h, A) in ((50000, 64, 16), (500000, 128, 32)):
    eng.generate_library(20261004, F, h, h, 0.0)
    eng.generate_patches(20261004, A)
    run("cw=0 %dx%d F=%d A=%d" % (h, h, F, A), eng, A, 200 if F < 100000 else 40)
# three and four chemicals: hue = k * (255 // n), S = 127 on grains (scripts/run_experiment.py:131,192)
rng = np.random.default_rng(1)
F, h, A = 20000, 64, 16
base = synth.synth_views(5, F, h, h)
for n_chem in (3, 4):
    lib = base.copy()
    lib[..., 0] = rng.integers(0, n_chem, lib.shape[:3]).astype(np.uint8) * (255 // n_chem)
    pats = synth.synth_patches(5, A, h, h)
    pats[..., 0] = rng.integers(0, n_chem, pats.shape[:3]).astype(np.uint8) * (255 // n_chem)
    for env, label in ((None, "timed"), ("1", "bytes forced")):
        if env: os.environ["DEJAVU_BITS"] = "0"
        e = navsim_amd.FamiliarityEngine(0)
        os.environ.pop("DEJAVU_BITS", None)
        e.set_library(lib, 0.25)
        e.upload_patches(pats)
        run("%d hues F=%d A=%d (%s)" % (n_chem, F, A, label), e, A)
        e.close()
