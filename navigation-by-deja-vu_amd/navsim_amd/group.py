"""FamiliarityGroup -- one process, several devices: the Python face of a dv_group (include/dejavu.h, csrc/dejavu_group.inl).

SURVEY 8-b1/b2 ask for `hip_sads_familiarity(chem_weight, devices=...)` over a context made from a list of device ids; the
reference is a single Python process (navsim/NavBySceneFamiliarity.py:72,140,299).  The library is cut into contiguous blocks of
views, one per member, every member scores its block, and the members' records are merged into the unsharded decision in C.
(The multi-process form -- one rank per GPU, one RCCL all-reduce per step -- is navsim_amd/sharded.py.)
"""
import ctypes

import numpy as np

from . import _native as N
from .engine import FamiliarityEngine


class FamiliarityGroup(object):
    """`devices`: the device id of each member (an id may repeat: independent contexts on one GPU)."""

    def __init__(self, devices):
        self._lib = N.load()
        devices = [int(d) for d in devices]
        if not devices:
            raise ValueError("no devices")
        ids = (ctypes.c_int * len(devices))(*devices)
        self._g = N._ctx_p()
        rc = self._lib.dv_group_create(ctypes.byref(self._g), ids, len(devices))
        if rc != 0:
            msg = self._lib.dv_last_error(None)
            raise N.EngineError("dv_group_create(%r) failed: %s (%s)" % (devices, msg.decode() if msg else "?", N.ERROR_NAMES.get(rc, rc)))
        self.devices = devices
        self.n_views = 0
        self.shape = None
        self.sensor_attached = False

    def _check(self, rc, what):
        if rc == 0:
            return
        msg = self._lib.dv_group_last_error(self._g)
        text = "%s failed: %s (%s)" % (what, msg.decode() if msg else "?", N.ERROR_NAMES.get(rc, rc))
        if rc == -5:
            raise IndexError(text)                                    # what the reference raises (util.pyx:137-168)
        if rc == -1:
            raise ValueError(text)
        raise N.EngineError(text)

    def close(self):
        if getattr(self, "_g", None):
            self._lib.dv_group_destroy(self._g)
            self._g = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __len__(self):
        return len(self.devices)

    def bounds(self, r):
        """(first view, one past the last) of member r's block."""
        ctx, first, n = N._ctx_p(), ctypes.c_int64(), ctypes.c_int64()
        self._check(self._lib.dv_group_member(self._g, int(r), ctypes.byref(ctx), ctypes.byref(first), ctypes.byref(n)), "dv_group_member")
        return first.value, first.value + n.value

    # -- library, landscape ---------------------------------------------------------------------
    def set_library(self, scenes, chem_weight=0.0):
        scenes = N.as_u8(scenes, "familiar_scenes")
        if scenes.ndim != 4:
            raise ValueError("familiar_scenes must be uint8[F,h,w,3]")
        F, h, w, ch = scenes.shape
        self._check(self._lib.dv_group_set_library(self._g, N.u8ptr(scenes), F, h, w, ch, float(chem_weight)), "dv_group_set_library")
        self.n_views, self.shape = F, (h, w)

    def set_landscape(self, landscape):
        landscape = N.as_u8(landscape, "landscape")
        self._check(self._lib.dv_group_set_landscape(self._g, N.u8ptr(landscape), landscape.shape[0], landscape.shape[1], 3),
                    "dv_group_set_landscape")

    def configure_sensor(self, sensor_dimensions, sensor_pixel_dimensions, lut, mask_middle_n):
        lut = np.ascontiguousarray(lut, dtype=np.uint8)
        assert lut.shape == (3, 256)
        self._check(self._lib.dv_group_configure_sensor(self._g, int(sensor_dimensions[0]), int(sensor_dimensions[1]),
                                                        int(sensor_pixel_dimensions[0]), int(sensor_pixel_dimensions[1]),
                                                        N.u8ptr(lut), int(mask_middle_n)), "dv_group_configure_sensor")

    def attach_sensor(self, landscape, sensor_dimensions, sensor_pixel_dimensions, lut, mask_middle_n):
        """Landscape and sensor model on every member (what navsim_amd.NavBySceneFamiliarity hands over after training): sense_step
        then needs nothing but the pose."""
        self.set_landscape(landscape)
        self.configure_sensor(sensor_dimensions, sensor_pixel_dimensions, lut, mask_middle_n)
        self.sensor_attached = True

    # -- scoring --------------------------------------------------------------------------------
    def score(self, scene, fambuf):
        """util.pyx:14-20 func(scene, fambuf): writes float64[F] in place, every member its block."""
        scene = N.as_u8(scene, "scene")
        if self.shape is None:
            raise N.EngineError("no library set")
        if tuple(scene.shape) != self.shape + (3,):
            raise ValueError("scene has shape %r, expected %r" % (tuple(scene.shape), self.shape + (3,)))
        if not (isinstance(fambuf, np.ndarray) and fambuf.dtype == np.float64):
            raise ValueError("Buffer dtype mismatch for fambuf, expected 'double'")
        if fambuf.shape != (self.n_views,):
            raise ValueError("fambuf has shape %r, expected (%d,)" % (fambuf.shape, self.n_views))
        buf = fambuf if fambuf.flags.c_contiguous else np.empty(self.n_views, dtype=np.float64)
        self._check(self._lib.dv_group_score(self._g, N.u8ptr(scene), N.f64ptr(buf)), "dv_group_score")
        if buf is not fambuf:
            fambuf[:] = buf
        return fambuf

    def _result(self, res, scene):
        d = FamiliarityEngine._result_dict(res, scene)
        d["resolved"] = bool(res.flags & N.DV_RES_RESOLVED)
        return d

    def step(self, patches, want_scene=True, force_resolve=False):
        """The heading loop of step_forward (:283-316) on patches uint8[A,h,w,3] over all members -> the dict engine.step returns."""
        patches = N.as_u8(patches, "patches")
        if patches.ndim != 4 or self.shape is None or tuple(patches.shape[1:]) != self.shape + (3,):
            raise ValueError("patches must be uint8[A,%s,3]" % (",".join(map(str, self.shape or ("h", "w"))),))
        res = N.StepResult()
        scene = np.empty(self.n_views, dtype=np.float64) if want_scene else None
        self._check(self._lib.dv_group_step(self._g, N.u8ptr(patches), patches.shape[0],
                                            N.DV_STEP_FORCE_RESOLVE if force_resolve else 0, ctypes.byref(res),
                                            N.f64ptr(scene) if want_scene else None), "dv_group_step")
        return self._result(res, scene)

    def sense_step(self, x, y, angles, want_scene=True, force_resolve=False):
        """One agent step's device work: every member senses the heading patches at (x, y) from its own copy of the landscape."""
        angles = np.ascontiguousarray(angles, dtype=np.float64).reshape(-1)
        res = N.StepResult()
        scene = np.empty(self.n_views, dtype=np.float64) if want_scene else None
        self._check(self._lib.dv_group_sense_step(self._g, float(x), float(y), N.f64ptr(angles), len(angles),
                                                  N.DV_STEP_FORCE_RESOLVE if force_resolve else 0, ctypes.byref(res),
                                                  N.f64ptr(scene) if want_scene else None), "dv_group_sense_step")
        return self._result(res, scene)
