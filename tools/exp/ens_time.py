#!/usr/bin/env python3
"""Timing-only driver of the ensemble passes (32 agents x 16 headings, 100 000 views of 64x64, patches uploaded): results are NOT
checked -- it is run under rocprofv3 with timing-experiment builds of the library (tools/runs/r4_lc22_exp.sh), whose sums are wrong
on purpose, and only the scoring kernel's duration is read from the trace."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "navigation-by-deja-vu_amd"))
import navsim_amd
from navsim_amd import synth
eng = navsim_amd.FamiliarityEngine(0)
eng.generate_library(20261004, 100000, 64, 64, chem_weight=0.25)
patches = synth.synth_patches(20261004, 32 * 16, 64, 64).reshape(32, 16, 64, 64, 3)
for _ in range(8):
    try:
        eng.step_batch(patches)
    except Exception as e:                      # (a timing build's sums may send a pass to paths that refuse them)
        print("step:", repr(e)[:100])
eng.close()
