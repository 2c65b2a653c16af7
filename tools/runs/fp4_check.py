"""fp4 form of the matrix-core kernel: does it engage, and are its sums the int8 form's?  (GPU run helper.)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "navigation-by-deja-vu_amd"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
from navsim_amd import synth
from navsim_amd.engine import Engine
os.environ["DEJAVU_SHAPE"] = "6"
os.environ["DEJAVU_BITS"] = "2"
for (F, h, w, A, cw) in ((5000, 32, 32, 32, 0.5), (3000, 20, 24, 7, 0.3), (70000, 16, 16, 64, 0.5), (5000, 32, 32, 16, 0.0)):
    lib = synth.synth_views(11, F, h, w)
    pat = synth.synth_patches(11, A, h, w)
    out = {}
    for fp4 in ("1", "0"):
        os.environ["DEJAVU_FP4"] = fp4
        eng = Engine()
        eng.set_library(lib, cw)
        fam = np.empty((A, F))
        for a in range(A):
            eng.score(pat[a], fam[a])
        r = eng.step(pat[:min(A, 64)], want_scene=A <= 32)
        info = eng.library_info()
        on = eng.patches_on_level() if info["fp4_form"] else None
        out[fp4] = fam.tobytes() + r["angle_familiarity"].tobytes() + r["angle_view"].tobytes()
        print(F, h, w, A, cw, "fp4 env", fp4, "fp4_form", info["fp4_form"], "planes", info["bit_planes_hs"], info["bit_planes_v"], "on level", on)
        eng.close()
    print("   identical:", out["1"] == out["0"])
