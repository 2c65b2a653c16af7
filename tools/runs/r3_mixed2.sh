cd $GRAFT_REPO_ROOT
python - <<'PY'
import time, numpy as np, sys, os
sys.path.insert(0, "navigation-by-deja-vu_amd"); sys.path.insert(0, ".")
import navsim_amd
for shape in ("0", "6", "5"):
    os.environ["DEJAVU_SHAPE"] = shape
    os.environ["DEJAVU_VERBOSE"] = "1"
    eng = navsim_amd.FamiliarityEngine(0)
    F, h, w, A, cw, seed = 500000, 128, 128, 32, 0.25, 20261004
    eng.generate_library(seed, F, h, w, cw, full_range_s=True)
    info = eng.library_info()
    eng.generate_patches(seed, A)
    for i in range(3):
        eng.step_enqueue(); r = eng.step_wait()
    eng.profile_kernel(True)
    t0 = time.perf_counter()
    for i in range(10):
        eng.generate_patches(seed + 5 + i, A); eng.step_enqueue(); r = eng.step_wait()
    dt = (time.perf_counter() - t0) / 10
    kms, kn = eng.profile_read()
    print("DEJAVU_SHAPE=%s: mixed_layout %s shape %d  step %.3f ms  kernel %.3f ms  form %s" % (shape, info["mixed_layout"], eng.workgroup_shape(A), dt * 1e3, kms / max(kn, 1), eng.scoring_form()))
    eng.close()
PY
