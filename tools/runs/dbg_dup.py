import os, sys, json
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "navigation-by-deja-vu_amd"))
import numpy as np
import navsim_amd
from tests.helpers import step_case_inputs
m = json.load(open(os.path.join(REPO, "tests/golden/manifest.json")))
cases = {c["name"]: c for c in m["t2_step"]}
for env in ({}, {"DEJAVU_BITS": "0"}, {"DEJAVU_FUSE": "0"}, {"DEJAVU_SHAPE": "1"}, {"DEJAVU_FINISH": "2"}, {"DEJAVU_FINISH": "0"}):
    os.environ.update(env)
    e = navsim_amd.FamiliarityEngine(0)
    for k in env: os.environ.pop(k)
    for name in ("s_dup", "s_dup"):
        lib, patches = step_case_inputs(cases[name])
        e.set_library(lib, cases[name]["chem_weight"])
        r = e.step(patches, want_scene=False)
        print(env, name, "shape", e.workgroup_shape(len(patches)), "n_cand", r["n_candidates"], "flags", r["flags"], "best", r["best_idex"], r["best_view"], flush=True)
    e.close()
