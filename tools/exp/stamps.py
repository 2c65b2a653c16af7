#!/usr/bin/env python3
"""Phase timing of the matrix-core scoring kernel from in-kernel wall-clock stamps (diagnostic build, never the product).

    python tools/exp/stamps.py build          # here: compiles csrc with -DDEJAVU_STAMPS into tools/exp/libdejavu_stamps.so
    python tools/exp/stamps.py run [F h A]    # on the GPU box: one workload, stamps of the last step, per-phase statistics
Stamps (100 MHz counter, per workgroup, first item only): 0 kernel entry, 1 first ring stage landed, 2 ring loop done,
3 before / 4 after the fused finishing, 5 exit."""
import ctypes, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CSRC = os.path.join(ROOT, "navigation-by-deja-vu_amd", "csrc")
SO = os.path.join(ROOT, "tools", "exp", os.environ.get("STAMPS_SO", "libdejavu_stamps.so"))
if sys.argv[1] == "build":         # build [extra -D flags]: e.g. STAMPS_SO=libdejavu_stamps_nocoef.so ... build -DDEJAVU_EXP_SKIP=1 (results wrong, timing only)
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-DDEJAVU_STAMPS"] + sys.argv[2:] +
                   ["-shared", "-o", SO, os.path.join(CSRC, "dejavu_hip.hip"), "-ldl"], check=True, cwd=CSRC)
    sys.exit(0)
sys.path.insert(0, os.path.join(ROOT, "navigation-by-deja-vu_amd"))
from navsim_amd import _native
_native.LIB_PATH = SO
import navsim_amd
F, h, A = (int(x) for x in sys.argv[2:5]) if len(sys.argv) >= 5 else (50000, 64, 16)
eng = navsim_amd.FamiliarityEngine(0)
eng.generate_library(20261004, F, h, h, 0.25)
for i in range(20):
    eng.generate_patches(100 + i, A)
    eng.step_enqueue(); eng.step_wait()
buf = (ctypes.c_ulonglong * (256 * 8))()
lib = _native.load()
lib.dv_debug_stamps.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
assert lib.dv_debug_stamps(eng._ctx, buf) == 0
st = np.frombuffer(buf, dtype=np.uint64).reshape(256, 8).astype(np.int64)
st = st[st[:, 0] > 0]
t0 = st[:, 0].min()
print("workgroups with stamps:", len(st), " form:", eng.scoring_form())
names = ["entry", "stage0 landed", "loop done", "before finish", "after finish", "exit"]
for i, n in enumerate(names):
    v = (st[:, i] - t0) / 100.0
    print("%-14s us after the first entry: min %7.2f  median %7.2f  max %7.2f" % (n, v.min(), np.median(v), v.max()))
for i in range(1, 6):
    d = (st[:, i] - st[:, i - 1]) / 100.0
    print("phase %d->%d: median %7.2f us  max %7.2f" % (i - 1, i, np.median(d), d.max()))
if os.environ.get("STAMPS_FIN"):          # a build with -DDEJAVU_EXP_FIN: slots 6, 7 = inside the first item's fused finishing
    for a, b, what in ((3, 6, "entries walked (this wave)"), (6, 7, "hand-over, item summary, thresholds (two barriers)"), (7, 4, "candidates listed, last barrier")):
        d = (st[:, b] - st[:, a]) / 100.0
        print("finishing, %-52s median %6.2f us  max %6.2f" % (what + ":", np.median(d), d.max()))
else:
    clk = (st[:, 7] - st[:, 6]) / np.maximum(st[:, 2] - st[:, 1], 1) * 100.0
    print("shader clock during the first item's loop: median %.0f MHz  min %.0f  max %.0f" % (np.median(clk), clk.min(), clk.max()))
eng.close()
