#!/usr/bin/env python3
"""Phase timing of k_sad_lc22 (ensemble passes of 64 headings) from in-kernel wall-clock stamps, diagnostic builds only:
    python tools/exp/stamps.py build [-DDEJAVU_EXP_FIN]      # here (STAMPS_SO names the library)
    python tools/exp/stamps22.py                             # on the GPU box
Stamps of a workgroup's FIRST item (100 MHz counter): 0 entry, 1 first stage landed, 2 loop + sums done, 3 / 4 in front of the first /
second heading tile's finishing, 5 behind both.  Without -DDEJAVU_EXP_FIN slots 6, 7 hold the shader clock counter at stamps 1, 2;
with it, the stamps inside the LAST finishing call (entries walked; second barrier passed)."""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SO = os.path.join(ROOT, "tools", "exp", os.environ.get("STAMPS_SO", "libdejavu_stamps.so"))
sys.path.insert(0, os.path.join(ROOT, "navigation-by-deja-vu_amd"))
from navsim_amd import _native
_native.LIB_PATH = SO
import navsim_amd
from navsim_amd import synth
os.environ.setdefault("DEJAVU_CHAINS", "1")            # one chain: the kernels run alone
eng = navsim_amd.FamiliarityEngine(0)
eng.generate_library(20261004, 100000, 64, 64, 0.25)
patches = synth.synth_patches(20261004, 8 * 16, 64, 64).reshape(8, 16, 64, 64, 3)        # two passes of 64 headings
for _ in range(6):
    eng.step_batch(patches)
buf = (ctypes.c_ulonglong * (256 * 8))()
lib = _native.load()
lib.dv_debug_stamps.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
assert lib.dv_debug_stamps(eng._ctx, buf) == 0
st = np.frombuffer(buf, dtype=np.uint64).reshape(256, 8).astype(np.int64)
st = st[st[:, 0] > 0]
t0 = st[:, 0].min()
print("workgroups with stamps:", len(st), " form:", eng.scoring_form())
names = ["entry", "stage0 landed", "loop+sums done", "before tile 0", "before tile 1", "after both"]
for i, n in enumerate(names):
    v = (st[:, i] - t0) / 100.0
    print("%-16s us after the first entry: min %7.2f  median %7.2f  max %7.2f" % (n, v.min(), np.median(v), v.max()))
for i in range(1, 6):
    d = (st[:, i] - st[:, i - 1]) / 100.0
    print("phase %d->%d: median %7.2f us  max %7.2f" % (i - 1, i, np.median(d), d.max()))
if os.environ.get("STAMPS_FIN"):
    for a, b, what in ((4, 6, "second call: entries walked (this wave)"), (6, 7, "second call: hand-over, item summary, thresholds (two barriers)"),
                       (7, 5, "second call: candidates listed, last barrier")):
        d = (st[:, b] - st[:, a]) / 100.0
        print("%-70s median %6.2f us  max %6.2f" % (what + ":", np.median(d), d.max()))
else:
    clk = (st[:, 7] - st[:, 6]) / np.maximum(st[:, 2] - st[:, 1], 1) * 100.0
    print("shader clock during the first item's loop: median %.0f MHz  min %.0f  max %.0f" % (np.median(clk), clk.min(), clk.max()))
eng.close()
