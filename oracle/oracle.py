"""ctypes front-end of oracle/liboracle.so -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Parity status: PINNED against the reference's own outputs (tests/golden/*.npz,
made by tests/golden/make_golden.py; checked by tests/test_oracle_golden.py).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.  Citations are into /root/reference/.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")
_lib = None

_u8p = ctypes.POINTER(ctypes.c_uint8)
_f64p = ctypes.POINTER(ctypes.c_double)
_i64p = ctypes.POINTER(ctypes.c_int64)
_i32p = ctypes.POINTER(ctypes.c_int32)


def build():
    """Compile liboracle.so with gcc (see oracle/Makefile)."""
    subprocess.run(["make", "-s", "-C", _HERE, "liboracle.so"], check=True)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = ctypes.CDLL(_LIB_PATH)
        L.oracle_sads_hsv.argtypes = [_u8p, ctypes.c_int64, ctypes.c_int, ctypes.c_int,
                                      _u8p, ctypes.c_double, _f64p]
        L.oracle_sads_hsv.restype = None
        L.oracle_step.argtypes = [_u8p, ctypes.c_int64, ctypes.c_int, ctypes.c_int,
                                  _u8p, ctypes.c_int, ctypes.c_double,
                                  _f64p, _f64p, _i32p, _i64p, _f64p]
        L.oracle_step.restype = ctypes.c_int
        L.oracle_ssds.argtypes = [_f64p, _f64p, ctypes.c_int64, ctypes.c_int64]
        L.oracle_ssds.restype = ctypes.c_double
        L.oracle_int_sums.argtypes = [_u8p, ctypes.c_int64, ctypes.c_int, ctypes.c_int,
                                      _u8p, _i64p, _i64p]
        L.oracle_int_sums.restype = None
        _lib = L
    return _lib


def _u8(a):
    a = np.ascontiguousarray(a)
    if a.dtype != np.uint8:
        # the reference rejects other dtypes with "Buffer dtype mismatch" (util.pyx:31-33)
        raise ValueError("Buffer dtype mismatch, expected 'uint8_t' but got %r" % (a.dtype,))
    return a


def sads_hsv(library, scene, chem_weight=0.0, fambuf=None):
    """navsim/util.pyx:31-73 -- fambuf[f] for one scene against every stored view."""
    library = _u8(library)
    scene = _u8(scene)
    F, h, w, c = library.shape
    assert c == 3 and scene.shape == (h, w, 3)
    if fambuf is None:
        fambuf = np.empty(F, dtype=np.float64)
    assert fambuf.dtype == np.float64 and fambuf.flags.c_contiguous and fambuf.shape == (F,)
    lib().oracle_sads_hsv(library.ctypes.data_as(_u8p), F, h, w,
                          scene.ctypes.data_as(_u8p), float(chem_weight),
                          fambuf.ctypes.data_as(_f64p))
    return fambuf


def sads_familiarity(chem_weight=0.0):
    """navsim/util.pyx:10-25 -- the two-stage plug-in factory, backed by the C restatement."""
    def sads_familiarity_internal(scenes):
        assert 0 <= chem_weight <= 1
        maxfam = scenes[0].shape[0] * scenes[0].shape[1]

        def func(scene, fambuf):
            sads_hsv(scenes, scene, chem_weight, fambuf)

        func.max_familiarity = maxfam
        return func
    return sads_familiarity_internal


def step(library, patches, chem_weight=0.0, want_scene=True):
    """navsim/NavBySceneFamiliarity.py:283-316 -- heading loop given the A patches.

    Returns dict(angle_familiarity, scene_familiarity, best_idex, best_view, step_familiarity).
    """
    library = _u8(library)
    patches = _u8(patches)
    F, h, w, _ = library.shape
    A = patches.shape[0]
    assert patches.shape == (A, h, w, 3)
    angle = np.empty(A, dtype=np.float64)
    scene = np.empty(F, dtype=np.float64) if want_scene else None
    bi = ctypes.c_int32(-1)
    bv = ctypes.c_int64(-1)
    sf = ctypes.c_double(0.0)
    rc = lib().oracle_step(library.ctypes.data_as(_u8p), F, h, w,
                           patches.ctypes.data_as(_u8p), A, float(chem_weight),
                           angle.ctypes.data_as(_f64p),
                           scene.ctypes.data_as(_f64p) if want_scene else None,
                           ctypes.byref(bi), ctypes.byref(bv), ctypes.byref(sf))
    if rc != 0:
        raise MemoryError("oracle_step scratch allocation failed")
    return dict(angle_familiarity=angle, scene_familiarity=scene, best_idex=int(bi.value),
                best_view=int(bv.value), step_familiarity=float(sf.value))


def ssds(a, b):
    """navsim/util.pyx:171-184."""
    a = np.ascontiguousarray(a, dtype=np.float64)
    b = np.ascontiguousarray(b, dtype=np.float64)
    assert a.shape == b.shape and a.ndim == 2
    return float(lib().oracle_ssds(a.ctypes.data_as(_f64p), b.ctypes.data_as(_f64p),
                                   a.shape[0], a.shape[1]))


def int_sums(library, scene):
    """Exact integer sums (S_hs, S_v) per view: util.pyx:48-56 and :69 without the weights."""
    library = _u8(library)
    scene = _u8(scene)
    F, h, w, _ = library.shape
    s_hs = np.empty(F, dtype=np.int64)
    s_v = np.empty(F, dtype=np.int64)
    lib().oracle_int_sums(library.ctypes.data_as(_u8p), F, h, w, scene.ctypes.data_as(_u8p),
                          s_hs.ctypes.data_as(_i64p), s_v.ctypes.data_as(_i64p))
    return s_hs, s_v


_omp = None


def _omp_lib():
    global _omp
    if _omp is None:
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "liboracle_omp.so")
        if not os.path.exists(path):
            raise RuntimeError("oracle/liboracle_omp.so is not built (make -C oracle)")
        _omp = ctypes.CDLL(path)
        _omp.oracle_step_fast.restype = ctypes.c_int
        _omp.oracle_step_fast.argtypes = [_u8p, ctypes.c_int64, ctypes.c_int, ctypes.c_int, _u8p, ctypes.c_int,
                                          ctypes.c_double, ctypes.c_int, _f64p, _i64p, ctypes.POINTER(ctypes.c_int32)]
        _omp.oracle_synth_int_sums.restype = ctypes.c_int
        _omp.oracle_synth_int_sums.argtypes = [ctypes.c_uint64, ctypes.c_int64, ctypes.c_int64, ctypes.c_int, ctypes.c_int,
                                               ctypes.c_int, _u8p, ctypes.c_int, _i64p, _i64p]
    return _omp


def have_omp():
    return os.path.exists(os.path.join(os.path.dirname(os.path.abspath(__file__)), "liboracle_omp.so"))


def synth_int_sums(seed, first_view, n_views, h, w, scene, full_range_s=False, threads=None):
    """int_sums of `scene` against views [first_view, first_view + n_views) of navsim_amd.synth.synth_views(seed, ...),
    regenerated view by view inside the C loop (dense checks of full-size synthetic libraries)."""
    scene = _u8(scene)
    assert scene.shape == (h, w, 3)
    s_hs = np.empty(n_views, dtype=np.int64)
    s_v = np.empty(n_views, dtype=np.int64)
    if threads is None:
        threads = max(1, min(32, os.cpu_count() or 1))
    _omp_lib().oracle_synth_int_sums(int(seed), int(first_view), int(n_views), int(h), int(w), 1 if full_range_s else 0,
                                     scene.ctypes.data_as(_u8p), int(threads), s_hs.ctypes.data_as(_i64p),
                                     s_v.ctypes.data_as(_i64p))
    return s_hs, s_v


def fam_from_sums(s_hs, s_v, n_px, chem_weight):
    """fam = P - (0.5 cw S_hs + (1 - cw) S_v) / 255 from the exact integer sums (util.pyx:59-73 with the per-pixel
    division taken out of the loop: within ~1e-12 of the sequential doubles)."""
    return float(n_px) - (0.5 * chem_weight * s_hs.astype(np.float64) + (1.0 - chem_weight) * s_v.astype(np.float64)) / 255.0


def step_fast(library, patches, chem_weight, threads):
    """Multi-core integer form of the step (oracle_step_fast in liboracle_omp.so); bench.py's second CPU figure only."""
    _omp = _omp_lib()
    library = _u8(library)
    patches = _u8(patches)
    F, h, w, _ = library.shape
    A = patches.shape[0]
    fam = np.empty(A, dtype=np.float64)
    view = np.empty(A, dtype=np.int64)
    best = ctypes.c_int32(0)
    _omp.oracle_step_fast(library.ctypes.data_as(_u8p), F, h, w, patches.ctypes.data_as(_u8p), A, float(chem_weight),
                          int(threads), fam.ctypes.data_as(_f64p), view.ctypes.data_as(_i64p), ctypes.byref(best))
    return dict(best_idex=int(best.value), best_view=int(view[best.value]), angle_familiarity=fam, angle_view=view)
