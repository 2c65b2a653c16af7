#!/usr/bin/env python3
"""Time of k_patch_prep alone (generator mode): usage prep_time.py <lib.so> F h A   (timing experiments; a truncated build leaves no valid patches)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "navigation-by-deja-vu_amd"))
from navsim_amd import _native
_native.LIB_PATH = os.path.abspath(sys.argv[1])
import navsim_amd
F, h, A = (int(x) for x in sys.argv[2:5])
eng = navsim_amd.FamiliarityEngine(0)
eng.generate_library(20261004, F, h, h, 0.25)
eng.generate_patches(1, A); eng.step_enqueue(); eng.step_wait()          # the engine settles on its scoring form
lib = _native.load()
n = 300
for i in range(20):
    eng.generate_patches(100 + i, A)
lib.dv_timer_start(eng._ctx)
for i in range(n):
    eng.generate_patches(200 + i, A)
import ctypes
ms = ctypes.c_float()
lib.dv_timer_stop.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_float)]
lib.dv_timer_stop(eng._ctx, ctypes.byref(ms))
print("%s: %d x %dx%d x %d headings: %.2f us per preparation (back to back, events)" % (os.path.basename(sys.argv[1]), F, h, h, A, ms.value / n * 1e3))
eng.close()
