"""FamiliarityEngine -- Python face of one dv_ctx (one GPU, one stored-view library shard)."""
import ctypes

import numpy as np

from . import _native as N


class BatchResults(object):
    """The records of an ensemble step (dv_step_batch / dv_sense_step_batch), one per agent.  A sequence of the per-agent result
    dictionaries step() returns -- made when asked for -- and, for callers that only move agents, the same numbers as arrays over
    the records' own memory: best_idex[n], best_view[n], step_familiarity[n], flags[n], n_candidates[n], angle_familiarity[n, A],
    angle_view[n, A].  (32 dictionaries per ensemble step cost the host 60-80 us beside a 0.8 ms device step.)"""
    _DTYPE = np.dtype(N.StepResult)

    def __init__(self, raw, n, A):
        self._raw, self.n, self.A = raw, n, A
        rec = np.frombuffer(raw, dtype=self._DTYPE, count=n)
        self.records = rec
        self.best_idex, self.best_view, self.step_familiarity = rec["best_heading"], rec["best_view"], rec["best_fam"]
        self.flags, self.n_candidates = rec["flags"], rec["n_candidates"]
        self.angle_familiarity, self.angle_view = rec["angle_fam"][:, :A], rec["angle_view"][:, :A]

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self[j] for j in range(*i.indices(self.n))]
        if i < 0:
            i += self.n
        if not 0 <= i < self.n:
            raise IndexError(i)
        return FamiliarityEngine._result_dict(self._raw[i], None)

    def __iter__(self):
        return (self[i] for i in range(self.n))


class FamiliarityEngine(object):
    """Scores sensor patches against a stored-view library resident in HBM.

    Replaces, behind the reference's interfaces, navsim/util.pyx:31-73 (the kernel) and the
    heading loop navsim/NavBySceneFamiliarity.py:283-316 (`step`).
    """

    def __init__(self, device=0, exact=False):
        self._lib = N.load()
        self._begun = False                    # an agent step was begun and not ended (agent_step_begin)
        self._ctx_raw = N._ctx_p()
        rc = self._lib.dv_create(ctypes.byref(self._ctx_raw), int(device))
        if rc != 0:
            msg = self._lib.dv_last_error(None)
            raise N.EngineError("dv_create(device=%d) failed: %s (%s)" % (
                device, msg.decode() if msg else "?", N.ERROR_NAMES.get(rc, rc)))
        self.device = int(device)
        self.n_views = 0
        self.shape = None
        self._step_state = None                  # sense_step_into's result record and angle buffer
        self._agent_state = None                 # agent_step's argument buffers
        self._err_out = ctypes.c_double()        # path_error_wait's answer
        self._err_out_ref = ctypes.byref(self._err_out)
        if exact:
            self.set_exact(True)

    # -- plumbing -------------------------------------------------------------------------------
    @property
    def _ctx(self):
        # Every call into the context goes through here, except agent_step_end: whatever it is, it supersedes an agent step that was
        # begun and not ended (the begun step's work stays harmlessly queued on the stream; its record is not read)
        self._begun = False
        return self._ctx_raw

    def _check(self, rc, what):
        if rc != 0:
            msg = self._lib.dv_last_error(self._ctx)
            text = "%s failed: %s (%s)" % (what, msg.decode() if msg else "?", N.ERROR_NAMES.get(rc, rc))
            if rc == -1:
                raise ValueError(text)
            raise N.EngineError(text)

    def close(self):
        if getattr(self, "_ctx_raw", None):
            self._lib.dv_destroy(self._ctx_raw)
            self._ctx_raw = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_exact(self, exact):
        """exact=True: every score is the reference's sequential-double value (fp64 kernel)."""
        self._check(self._lib.dv_set_exact(self._ctx, 1 if exact else 0), "dv_set_exact")

    def set_stream(self, hip_stream):
        self._check(self._lib.dv_set_stream(self._ctx, ctypes.c_void_p(hip_stream or 0)), "dv_set_stream")

    def synchronize(self):
        self._check(self._lib.dv_synchronize(self._ctx), "dv_synchronize")

    # -- library --------------------------------------------------------------------------------
    def set_library(self, scenes, chem_weight=0.0, first_view=0):
        """scenes: uint8[F,h,w,3] (NavBySceneFamiliarity.py:122).  Copied to the GPU and re-tiled."""
        scenes = N.as_u8(scenes, "familiar_scenes")
        if scenes.ndim != 4:
            raise ValueError("familiar_scenes must be uint8[F,h,w,3], got shape %r" % (scenes.shape,))
        F, h, w, ch = scenes.shape
        self._check(self._lib.dv_set_library(self._ctx, N.u8ptr(scenes), F, h, w, ch, float(chem_weight),
                                             int(first_view)), "dv_set_library")
        self.n_views, self.shape = F, (h, w)

    def generate_library(self, seed, n_views, h, w, chem_weight=0.0, first_view=0, full_range_s=False):
        """Same bytes as synth.synth_views(seed, n_views, h, w, first_view, full_range_s), generated in HBM."""
        self._check(self._lib.dv_generate_library_ex(self._ctx, int(seed), int(n_views), int(h), int(w), float(chem_weight),
                                                     int(first_view), 1 if full_range_s else 0), "dv_generate_library_ex")
        self.n_views, self.shape = int(n_views), (int(h), int(w))

    def append_library(self, scenes):
        """More views (uint8[n,h,w,3]) behind the resident ones; only the new view groups are re-tiled.  Raises
        EngineError (DV_ERR_STATE) when they do not fit the resident layout (a new hue, S > 127 on a signed plane)."""
        scenes = N.as_u8(scenes, "familiar_scenes")
        if scenes.ndim != 4 or scenes.shape[1:3] != tuple(self.shape):
            raise ValueError("appended views must be uint8[n,%d,%d,3], got shape %r" % (self.shape + (scenes.shape,)))
        self._check(self._lib.dv_append_library(self._ctx, N.u8ptr(scenes), scenes.shape[0], scenes.shape[3]), "dv_append_library")
        self.n_views += scenes.shape[0]

    def append_library_from_poses(self, x, y, angle, want_views=True):
        """dv_append_library with the views sensed on the device; returns them (uint8[n,h,w,3]) when want_views."""
        x, y, angle = self._pose_arrays(x, y, angle)
        h, w = self.sensor_shape
        out = np.empty((len(x), h, w, 3), dtype=np.uint8) if want_views else None
        self._check_sense(self._lib.dv_append_library_from_poses(self._ctx, N.f64ptr(x), N.f64ptr(y), N.f64ptr(angle), len(x),
                                                                 N.u8ptr(out) if want_views else None), "dv_append_library_from_poses")
        self.n_views += len(x)
        return out

    def clear_library(self):
        self._check(self._lib.dv_clear_library(self._ctx), "dv_clear_library")
        self.n_views, self.shape = 0, None

    def library_info(self):
        info = N.LibInfo()
        self._check(self._lib.dv_get_library_info(self._ctx, ctypes.byref(info)), "dv_get_library_info")
        return dict(n_views=info.n_views, first_view=info.first_view, h=info.h, w=info.w,
                    n_planes=info.n_planes, n_hue_planes=info.n_hue_planes, generic_hue=bool(info.generic_hue),
                    has_value_plane=bool(info.has_value_plane), tile_bytes=info.tile_bytes,
                    chem_weight=info.chem_weight, delta=info.delta, hues=[int(x) for x in info.hues][:info.n_hues],
                    signed_saturation=bool(info.signed_saturation), has_bit_planes=bool(info.has_bit_planes),
                    bit_planes_hs=info.bit_planes_hs, bit_planes_v=info.bit_planes_v, bit_tile_bytes=info.bit_tile_bytes,
                    fp4_form=bool(info.fp4_form), code_tile_bytes=info.code_tile_bytes, mixed_layout=bool(info.mixed_layout))

    def read_planes(self, v0, n):
        info = self.library_info()
        out = np.empty((n, info["n_planes"], info["h"] * info["w"]), dtype=np.uint8)
        self._check(self._lib.dv_read_planes(self._ctx, int(v0), int(n), N.u8ptr(out)), "dv_read_planes")
        return out

    # -- sensor model on the GPU ---------------------------------------------------------------
    def set_landscape(self, landscape):
        """landscape: uint8[rows, cols, 3] HSV (any strides; copied).  Kept resident in HBM."""
        landscape = N.as_u8(landscape, "landscape")
        if landscape.ndim != 3 or landscape.shape[2] != 3:
            raise ValueError("landscape must be uint8[rows, cols, 3], got shape %r" % (landscape.shape,))
        self._check(self._lib.dv_set_landscape(self._ctx, N.u8ptr(landscape), landscape.shape[0], landscape.shape[1], 3),
                    "dv_set_landscape")

    def configure_sensor(self, sensor_dimensions, sensor_pixel_dimensions, lut, mask_middle_n):
        lut = np.ascontiguousarray(lut, dtype=np.uint8)
        assert lut.shape == (3, 256)
        self._check(self._lib.dv_configure_sensor(self._ctx, int(sensor_dimensions[0]), int(sensor_dimensions[1]),
                                                  int(sensor_pixel_dimensions[0]), int(sensor_pixel_dimensions[1]),
                                                  N.u8ptr(lut), int(mask_middle_n)), "dv_configure_sensor")
        self.sensor_shape = (int(sensor_dimensions[1]), int(sensor_dimensions[0]))

    @staticmethod
    def _pose_arrays(x, y, angle):
        x = np.ascontiguousarray(x, dtype=np.float64).reshape(-1)
        y = np.ascontiguousarray(y, dtype=np.float64).reshape(-1)
        angle = np.ascontiguousarray(angle, dtype=np.float64).reshape(-1)
        assert x.shape == y.shape == angle.shape
        return x, y, angle

    def _check_sense(self, rc, what):
        if rc == -5:
            msg = self._lib.dv_last_error(self._ctx)
            raise IndexError(msg.decode() if msg else "index out of bounds")     # what the reference raises
        self._check(rc, what)

    def sense(self, x, y, angle):
        """get_sensor_mat at n poses -> uint8[n, h, w, 3] (NavBySceneFamiliarity.py:151-192)."""
        x, y, angle = self._pose_arrays(x, y, angle)
        h, w = self.sensor_shape
        out = np.empty((len(x), h, w, 3), dtype=np.uint8)
        self._check_sense(self._lib.dv_sense(self._ctx, N.f64ptr(x), N.f64ptr(y), N.f64ptr(angle), len(x), N.u8ptr(out)),
                          "dv_sense")
        return out

    def sense_patches(self, x, y, angles):
        """The heading patches of one position, sensed straight into the resident patches."""
        angles = np.ascontiguousarray(angles, dtype=np.float64).reshape(-1)
        self._check_sense(self._lib.dv_sense_patches(self._ctx, float(x), float(y), N.f64ptr(angles), len(angles)),
                          "dv_sense_patches")

    def sense_step(self, x, y, angles, want_scene=True, force_resolve=False):
        """Sense the heading patches at (x, y) and score them: one call for an agent step's device work."""
        angles = np.ascontiguousarray(angles, dtype=np.float64).reshape(-1)
        if len(angles) > N.DV_MAX_HEADINGS:
            return self._wide(lambda flags, res, fam, view, scene: self._lib.dv_sense_step_wide(
                self._ctx, float(x), float(y), N.f64ptr(angles), len(angles), flags, res, fam, view, scene),
                len(angles), want_scene, force_resolve, "dv_sense_step_wide")
        r = N.StepResult()
        scene = np.empty(self.n_views, dtype=np.float64) if want_scene else None
        self._check_sense(self._lib.dv_sense_step(self._ctx, float(x), float(y), N.f64ptr(angles), len(angles),
                                                  N.DV_STEP_FORCE_RESOLVE if force_resolve else 0, ctypes.byref(r),
                                                  N.f64ptr(scene) if want_scene else None), "dv_sense_step")
        return self._result_dict(r, scene)

    def sense_step_into(self, x, y, angle, offsets, out_fam):
        """The agent's step as it runs thousands of times per second: headings (angle + offsets) mod 2 pi, sensed and scored,
        per-heading familiarities written into out_fam; returns the chosen heading's index.  Same call as sense_step(...,
        want_scene=False) -- dv_sense_step -- through ONE result record and ONE angle buffer kept for the engine's lifetime
        (no per-step allocations, views or dictionaries on the host side)."""
        if len(offsets) > N.DV_MAX_HEADINGS:
            # more headings than one library pass holds: the wide step (passes merged in the library, same decision rule)
            res = self.sense_step(x, y, (angle + offsets) % (2 * np.pi), want_scene=False)
            out_fam[:] = res["angle_familiarity"]
            return res["best_idex"]
        if out_fam.shape != (len(offsets),):
            raise ValueError("out_fam has shape %r, expected (%d,)" % (out_fam.shape, len(offsets)))
        st = self._step_state
        if st is None or len(st[0]) != len(offsets):
            res = N.StepResult()
            buf = np.empty(len(offsets), dtype=np.float64)
            st = self._step_state = (buf, N.f64ptr(buf), res, ctypes.byref(res),
                                     np.frombuffer(res, dtype=np.float64, count=len(offsets), offset=N.RESULT_ARRAYS_OFFSET))
        buf, bufp, res, resp, fam = st
        np.add(offsets, angle, out=buf)
        np.mod(buf, 2 * np.pi, out=buf)
        rc = self._lib.dv_sense_step(self._ctx, x, y, bufp, len(buf), 0, resp, None)
        if rc:
            self._check_sense(rc, "dv_sense_step")
        out_fam[:] = fam
        return res.best_heading

    def _agent_args(self, offsets, out_fam):
        st = self._agent_state
        if st is None or st[0] is not offsets or st[1] is not out_fam:
            if not (out_fam.flags.c_contiguous and out_fam.dtype == np.float64 and out_fam.shape == (len(offsets),)):
                raise ValueError("out_fam must be a C-contiguous float64[%d]" % len(offsets))
            off = np.ascontiguousarray(offsets, dtype=np.float64)
            best, nearest, have = ctypes.c_int32(0), ctypes.c_double(0.0), ctypes.c_int32(0)
            st = self._agent_state = (offsets, out_fam, off, N.f64ptr(off), len(off), N.f64ptr(out_fam), best, ctypes.byref(best),
                                      nearest, ctypes.byref(nearest), have, ctypes.byref(have))
        return st

    def agent_step(self, x, y, angle, offsets, out_fam, error_pos, reach):
        """dv_agent_step: one agent step's device work and device-side book-keeping in one call.  The headings (angle + offsets)
        mod 2 pi are sensed at (x, y) and scored, out_fam (float64[A], C-contiguous: the agent's angle_familiarity) is written in
        place; error_pos = (ex, ey) asks for the error metrics of that position (or None), and an outstanding answer comes back.
        Returns (best heading, nearest distance or None)."""
        _, _, _, offp, A, famp, best, bestp, nearest, nearestp, have, havep = self._agent_args(offsets, out_fam)
        if error_pos is None:
            rc = self._lib.dv_agent_step(self._ctx, x, y, angle, offp, A, 0, 0.0, 0.0, 0.0, famp, bestp, nearestp, havep)
        else:
            rc = self._lib.dv_agent_step(self._ctx, x, y, angle, offp, A, 1, error_pos[0], error_pos[1], reach, famp, bestp, nearestp, havep)
        if rc:
            self._check_sense(rc, "dv_agent_step")
        return best.value, (nearest.value if have.value else None)

    def agent_step_begin(self, x, y, angle, offsets, out_fam, error_pos, reach):
        """First half of agent_step (dv_agent_step_begin): collects an outstanding error answer, launches the step, returns at once
        with that answer (or None).  agent_step_end() hands out the step's result -- unless anything else was asked of the engine in
        between, which supersedes the begun step."""
        _, _, _, offp, A, famp, best, bestp, nearest, nearestp, have, havep = self._agent_args(offsets, out_fam)
        if error_pos is None:
            rc = self._lib.dv_agent_step_begin(self._ctx, x, y, angle, offp, A, 0, 0.0, 0.0, 0.0, nearestp, havep)
        else:
            rc = self._lib.dv_agent_step_begin(self._ctx, x, y, angle, offp, A, 1, error_pos[0], error_pos[1], reach, nearestp, havep)
        if rc:
            self._check_sense(rc, "dv_agent_step_begin")
        self._begun = True
        return nearest.value if have.value else None

    def agent_step_end(self):
        """Second half: waits for the begun step, writes its per-heading maxima into the out_fam given to agent_step_begin and
        returns the best heading; None when the begun step was superseded (the caller takes the step again)."""
        if not self._begun:
            return None
        self._begun = False
        st = self._agent_state
        rc = self._lib.dv_agent_step_end(self._ctx_raw, st[5], st[7])
        if rc:
            self._check_sense(rc, "dv_agent_step_end")
        return st[6].value

    def agent_step_end_begin(self, cand, bounds, do_error, reach):
        """agent_step_end and the next agent_step_begin in one call (dv_agent_step_end_begin): cand = (angles[A], xs[A], ys[A]) of every
        candidate heading (float64, C-contiguous), bounds = float64[3] of the reference's bounds test.  Returns None when the begun step
        was superseded, else (best heading, whether the next step was begun, an outstanding error answer or None)."""
        if not self._begun:
            return None
        self._begun = False
        st = self._agent_state
        begun = ctypes.c_int32(0)
        rc = self._lib.dv_agent_step_end_begin(self._ctx_raw, st[5], st[7], N.f64ptr(cand[1]), N.f64ptr(cand[2]), N.f64ptr(cand[0]), st[3], st[4],
                                               N.f64ptr(bounds), 1 if do_error else 0, reach, ctypes.byref(begun), st[9], st[11])
        if rc:
            self._check_sense(rc, "dv_agent_step_end_begin")
        self._begun = bool(begun.value)
        return st[6].value, self._begun, (st[8].value if st[10].value else None)

    def sense_step_batch(self, x, y, angles, force_resolve=False):
        """Ensemble step on the device: agent i at (x[i], y[i]) looking along angles[i][0..A) -> BatchResults (a sequence of result dicts)."""
        x = np.ascontiguousarray(x, dtype=np.float64).reshape(-1)
        y = np.ascontiguousarray(y, dtype=np.float64).reshape(-1)
        angles = np.ascontiguousarray(angles, dtype=np.float64)
        if angles.ndim != 2 or len(x) != len(y) or angles.shape[0] != len(x):
            raise ValueError("x[N], y[N] and angles[N, A] expected")
        n, A = angles.shape
        res = (N.StepResult * n)()
        self._check_sense(self._lib.dv_sense_step_batch(self._ctx, N.f64ptr(x), N.f64ptr(y), N.f64ptr(angles), n, A,
                                                        N.DV_STEP_FORCE_RESOLVE if force_resolve else 0, res),
                          "dv_sense_step_batch")
        return BatchResults(res, n, A)

    def set_library_from_poses(self, x, y, angle, chem_weight=0.0, first_view=0, want_views=True):
        """train_from_path on the device: sense the poses and ingest them as the library; returns familiar_scenes."""
        x, y, angle = self._pose_arrays(x, y, angle)
        h, w = self.sensor_shape
        views = np.empty((len(x), h, w, 3), dtype=np.uint8) if want_views else None
        self._check_sense(self._lib.dv_set_library_from_poses(self._ctx, N.f64ptr(x), N.f64ptr(y), N.f64ptr(angle), len(x),
                                                              float(chem_weight), int(first_view),
                                                              N.u8ptr(views) if want_views else None),
                          "dv_set_library_from_poses")
        self.n_views, self.shape = len(x), (h, w)
        return views

    # -- scoring --------------------------------------------------------------------------------
    def _patch_shape_ok(self, p, lead):
        h, w = self.shape if self.shape else (None, None)
        want = lead + (h, w, 3)
        if self.shape is None:
            raise N.EngineError("no library set")
        if tuple(p.shape) != want:
            raise ValueError("patch array has shape %r, expected %r" % (tuple(p.shape), want))

    def score(self, scene, fambuf):
        """util.pyx:14-20 func(scene, fambuf): writes float64[F] in place."""
        scene = N.as_u8(scene, "scene")
        self._patch_shape_ok(scene, ())
        if not (isinstance(fambuf, np.ndarray) and fambuf.dtype == np.float64):
            raise ValueError("Buffer dtype mismatch for fambuf, expected 'double'")
        if fambuf.shape != (self.n_views,):
            raise ValueError("fambuf has shape %r, expected (%d,)" % (fambuf.shape, self.n_views))
        if fambuf.flags.c_contiguous:
            self._check(self._lib.dv_score(self._ctx, N.u8ptr(scene), N.f64ptr(fambuf)), "dv_score")
        else:
            tmp = np.empty(self.n_views, dtype=np.float64)
            self._check(self._lib.dv_score(self._ctx, N.u8ptr(scene), N.f64ptr(tmp)), "dv_score")
            fambuf[:] = tmp
        return fambuf

    @staticmethod
    def _result_dict(r, scene):
        A = r.n_headings
        # views of the record's four per-heading arrays ([4][64] 8-byte values after the header); every step
        # has its own record, which the views keep alive
        M = N.DV_MAX_HEADINGS
        f64 = np.frombuffer(r, dtype=np.float64, count=4 * M, offset=N.RESULT_ARRAYS_OFFSET)
        i64 = f64.view(np.int64)
        return dict(best_idex=r.best_heading, best_view=r.best_view, step_familiarity=r.best_fam,
                    angle_familiarity=f64[:A], angle_view=i64[M:M + A],
                    exact_familiarity=f64[2 * M:2 * M + A], exact_view=i64[3 * M:3 * M + A],
                    approx_max=r.approx_max, delta=r.delta, n_candidates=r.n_candidates,
                    flags=r.flags, scene_familiarity=scene)

    def step(self, patches, want_scene=True, force_resolve=False):
        """Heading loop of step_forward (:283-316) on patches uint8[A,h,w,3]."""
        patches = N.as_u8(patches, "patches")
        if patches.ndim != 4:
            raise ValueError("patches must be uint8[A,h,w,3]")
        self._patch_shape_ok(patches, (patches.shape[0],))
        if patches.shape[0] > N.DV_MAX_HEADINGS:
            return self._wide(lambda flags, res, fam, view, scene: self._lib.dv_step_wide(
                self._ctx, N.u8ptr(patches), patches.shape[0], flags, res, fam, view, scene),
                patches.shape[0], want_scene, force_resolve, "dv_step_wide")
        r = N.StepResult()
        scene = np.empty(self.n_views, dtype=np.float64) if want_scene else None
        self._check(self._lib.dv_step(self._ctx, N.u8ptr(patches), patches.shape[0],
                                      N.DV_STEP_FORCE_RESOLVE if force_resolve else 0, ctypes.byref(r),
                                      N.f64ptr(scene) if want_scene else None), "dv_step")
        return self._result_dict(r, scene)

    def _wide(self, call, A, want_scene, force_resolve, what):
        """A step of more than DV_MAX_HEADINGS headings (the reference takes any n_test_angles, NavBySceneFamiliarity.py:62,289):
        ceil(A / 64) library passes merged inside the library; same keys as a single-pass step's result."""
        res = N.WideResult()
        fam = np.empty(A, dtype=np.float64)
        view = np.empty(A, dtype=np.int64)
        scene = np.empty(self.n_views, dtype=np.float64) if want_scene else None
        self._check_sense(call(N.DV_STEP_FORCE_RESOLVE if force_resolve else 0, ctypes.byref(res), N.f64ptr(fam), N.i64ptr(view),
                               N.f64ptr(scene) if want_scene else None), what)
        return dict(best_idex=res.best_heading, best_view=res.best_view, step_familiarity=res.best_fam, angle_familiarity=fam,
                    angle_view=view, flags=res.flags, n_passes=res.n_passes, n_contending=res.n_contending,
                    scene_familiarity=scene)

    def step_batch(self, patches, force_resolve=False):
        """Ensemble step: patches uint8[N, A, h, w, 3] -> BatchResults, a sequence of N result dicts (one library pass per 64/A agents)."""
        patches = N.as_u8(patches, "patches")
        if patches.ndim != 5:
            raise ValueError("patches must be uint8[N,A,h,w,3]")
        n, A = patches.shape[0], patches.shape[1]
        self._patch_shape_ok(patches, (n, A))
        res = (N.StepResult * n)()
        self._check(self._lib.dv_step_batch(self._ctx, N.u8ptr(patches), n, A,
                                            N.DV_STEP_FORCE_RESOLVE if force_resolve else 0, res), "dv_step_batch")
        return BatchResults(res, n, A)

    # -- ssd_f32 metric --------------------------------------------------------------------------
    @staticmethod
    def _f32(a, what):
        a = np.ascontiguousarray(a)
        if a.dtype != np.float32:
            raise ValueError("Buffer dtype mismatch for %s, expected 'float' but got '%s'" % (what, a.dtype))
        return a

    def set_library_f32(self, views, first_view=0):
        """views: float32[F,h,w]; scores are sums of squared differences (navsim/util.pyx:171-184)."""
        views = self._f32(views, "views")
        if views.ndim != 3:
            raise ValueError("views must be float32[F,h,w], got shape %r" % (views.shape,))
        F, h, w = views.shape
        fp = views.ctypes.data_as(ctypes.POINTER(ctypes.c_float))
        self._check(self._lib.dv_set_library_f32(self._ctx, fp, F, h, w, int(first_view)), "dv_set_library_f32")
        self.n_views, self.shape = F, (h, w)

    def generate_library_f32(self, seed, n_views, h, w, first_view=0):
        """Same values as synth.synth_views_f32(seed, n_views, h, w, first_view), generated in HBM."""
        self._check(self._lib.dv_generate_library_f32(self._ctx, int(seed), int(n_views), int(h), int(w), int(first_view)),
                    "dv_generate_library_f32")
        self.n_views, self.shape = int(n_views), (int(h), int(w))

    def score_f32(self, patch, ssdbuf=None):
        patch = self._f32(patch, "patch")
        if tuple(patch.shape) != self.shape:
            raise ValueError("patch has shape %r, expected %r" % (patch.shape, self.shape))
        if ssdbuf is None:
            ssdbuf = np.empty(self.n_views, dtype=np.float64)
        self._check(self._lib.dv_score_f32(self._ctx, patch.ctypes.data_as(ctypes.POINTER(ctypes.c_float)),
                                           N.f64ptr(ssdbuf)), "dv_score_f32")
        return ssdbuf

    def step_f32(self, patches, want_scene=False, force_resolve=False):
        """Least-SSD heading over float32 patches [A,h,w]: dict with angle_ssd, best_idex, best_view, step_ssd."""
        patches = self._f32(patches, "patches")
        if patches.ndim != 3 or tuple(patches.shape[1:]) != self.shape:
            raise ValueError("patches must be float32[A,%d,%d]" % self.shape)
        r = N.StepResult()
        scene = np.empty(self.n_views, dtype=np.float64) if want_scene else None
        self._check(self._lib.dv_step_f32(self._ctx, patches.ctypes.data_as(ctypes.POINTER(ctypes.c_float)),
                                          patches.shape[0], N.DV_STEP_FORCE_RESOLVE if force_resolve else 0,
                                          ctypes.byref(r), N.f64ptr(scene) if want_scene else None), "dv_step_f32")
        d = self._result_dict(r, scene)
        return dict(best_idex=d["best_idex"], best_view=d["best_view"], step_ssd=d["step_familiarity"],
                    angle_ssd=d["angle_familiarity"], angle_view=d["angle_view"], n_candidates=d["n_candidates"],
                    flags=d["flags"], scene_ssd=scene)

    # -- ssd_u8 metric: exact SSD of uint8 views on the int8 matrix cores -------------------------------
    def set_library_u8(self, views, first_view=0):
        """views: uint8[F,h,w]; scores are the exact integer sums of squared differences (navsim/util.pyx:171-184 on uint8 data)."""
        views = N.as_u8(views, "views")
        if views.ndim != 3:
            raise ValueError("views must be uint8[F,h,w], got shape %r" % (views.shape,))
        F, h, w = views.shape
        self._check(self._lib.dv_set_library_u8(self._ctx, N.u8ptr(views), F, h, w, int(first_view)), "dv_set_library_u8")
        self.n_views, self.shape = F, (h, w)

    def set_library_u8_from_poses(self, x, y, angle, channel=2, first_view=0, want_views=True):
        """train_from_path for the ssd_u8 plug-in on the device: sense the poses, ingest their `channel` bytes as the library;
        returns familiar_scenes (uint8[n,h,w,3]) when want_views."""
        x, y, angle = self._pose_arrays(x, y, angle)
        h, w = self.sensor_shape
        views = np.empty((len(x), h, w, 3), dtype=np.uint8) if want_views else None
        self._check_sense(self._lib.dv_set_library_u8_from_poses(self._ctx, N.f64ptr(x), N.f64ptr(y), N.f64ptr(angle), len(x),
                                                                 int(channel), int(first_view),
                                                                 N.u8ptr(views) if want_views else None),
                          "dv_set_library_u8_from_poses")
        self.n_views, self.shape = len(x), (h, w)
        return views

    def sense_step_u8(self, x, y, angles, channel=2, want_scene=False):
        """dv_sense_step for an ssd_u8 library: the heading patches sensed at (x, y), their `channel` bytes scored on the int8
        matrix cores, the least-SSD heading decided on the device.  Result as step_u8's."""
        angles = np.ascontiguousarray(angles, dtype=np.float64).reshape(-1)
        r = N.StepResult()
        scene = np.empty(self.n_views, dtype=np.float64) if want_scene else None
        self._check_sense(self._lib.dv_sense_step_u8(self._ctx, float(x), float(y), N.f64ptr(angles), len(angles), int(channel), 0,
                                                     ctypes.byref(r), N.f64ptr(scene) if want_scene else None), "dv_sense_step_u8")
        d = self._result_dict(r, scene)
        return dict(best_idex=d["best_idex"], best_view=d["best_view"], step_ssd=d["step_familiarity"],
                    angle_ssd=d["angle_familiarity"], angle_view=d["angle_view"], n_candidates=d["n_candidates"],
                    flags=d["flags"], scene_ssd=scene)

    def score_u8(self, patch, ssdbuf=None):
        patch = N.as_u8(patch, "patch")
        if tuple(patch.shape) != self.shape:
            raise ValueError("patch has shape %r, expected %r" % (patch.shape, self.shape))
        if ssdbuf is None:
            ssdbuf = np.empty(self.n_views, dtype=np.float64)
        self._check(self._lib.dv_score_u8(self._ctx, N.u8ptr(patch), N.f64ptr(ssdbuf)), "dv_score_u8")
        return ssdbuf

    def step_u8(self, patches, want_scene=False):
        """Least-SSD heading over uint8 patches [A,h,w]: dict with angle_ssd, best_idex, best_view, step_ssd (exact integers)."""
        patches = N.as_u8(patches, "patches")
        if patches.ndim != 3 or tuple(patches.shape[1:]) != self.shape:
            raise ValueError("patches must be uint8[A,%d,%d]" % self.shape)
        r = N.StepResult()
        scene = np.empty(self.n_views, dtype=np.float64) if want_scene else None
        self._check(self._lib.dv_step_u8(self._ctx, N.u8ptr(patches), patches.shape[0], 0, ctypes.byref(r),
                                         N.f64ptr(scene) if want_scene else None), "dv_step_u8")
        d = self._result_dict(r, scene)
        return dict(best_idex=d["best_idex"], best_view=d["best_view"], step_ssd=d["step_familiarity"],
                    angle_ssd=d["angle_familiarity"], angle_view=d["angle_view"], n_candidates=d["n_candidates"],
                    flags=d["flags"], scene_ssd=scene)

    def resolve(self):
        r = N.StepResult()
        self._check(self._lib.dv_resolve(self._ctx, ctypes.byref(r)), "dv_resolve")
        return self._result_dict(r, None)

    # -- resident form --------------------------------------------------------------------------
    def upload_patches(self, patches):
        patches = N.as_u8(patches, "patches")
        self._patch_shape_ok(patches, (patches.shape[0],))
        self._check(self._lib.dv_upload_patches(self._ctx, N.u8ptr(patches), patches.shape[0]), "dv_upload_patches")

    def generate_patches(self, seed, n_headings):
        self._check(self._lib.dv_generate_patches(self._ctx, int(seed), int(n_headings)), "dv_generate_patches")

    def step_enqueue(self, want_scene=False, force_resolve=False):
        flags = (N.DV_STEP_WANT_SCENE if want_scene else 0) | (N.DV_STEP_FORCE_RESOLVE if force_resolve else 0)
        self._check(self._lib.dv_step_enqueue(self._ctx, flags), "dv_step_enqueue")

    def step_wait(self, want_scene=False):
        r = N.StepResult()
        scene = np.empty(self.n_views, dtype=np.float64) if want_scene else None
        self._check_sense(self._lib.dv_step_wait(self._ctx, ctypes.byref(r), N.f64ptr(scene) if want_scene else None),
                          "dv_step_wait")
        return self._result_dict(r, scene)

    def step_record(self):
        """(device pointer, n_doubles) of the packed record of the last enqueued step (sharded exchange)."""
        ptr = ctypes.c_void_p()
        n = ctypes.c_int(0)
        self._check(self._lib.dv_step_record(self._ctx, ctypes.byref(ptr), ctypes.byref(n)), "dv_step_record")
        return int(ptr.value), int(n.value)

    def resolve_enqueue(self):
        self._check(self._lib.dv_resolve_enqueue(self._ctx), "dv_resolve_enqueue")

    def step_keys(self, rank, world, signed_order=False):
        """Enqueue the packing of the last step's keys for the all-reduce(max) exchange; (device pointer, n_words)."""
        ptr = ctypes.c_void_p()
        n = ctypes.c_int(0)
        self._check(self._lib.dv_step_keys(self._ctx, int(rank), int(world), 1 if signed_order else 0, ctypes.byref(ptr),
                                           ctypes.byref(n)), "dv_step_keys")
        return int(ptr.value), int(n.value)

    # -- measurement ----------------------------------------------------------------------------
    def timer_start(self):
        self._check(self._lib.dv_timer_start(self._ctx), "dv_timer_start")

    def timer_stop(self):
        ms = ctypes.c_float(0)
        self._check(self._lib.dv_timer_stop(self._ctx, ctypes.byref(ms)), "dv_timer_stop")
        return float(ms.value)

    def profile_kernel(self, enable, every=1):
        """Bracket every `every`-th scoring-kernel launch with hipEvents (an event pair costs stream time)."""
        self._check(self._lib.dv_profile_kernel(self._ctx, max(1, int(every)) if enable else 0), "dv_profile_kernel")

    def profile_read(self):
        tot = ctypes.c_double(0)
        n = ctypes.c_int64(0)
        self._check(self._lib.dv_profile_read(self._ctx, ctypes.byref(tot), ctypes.byref(n)), "dv_profile_read")
        return float(tot.value), int(n.value)

    def publish(self, device_ptr, n_doubles):
        """Enqueue the hand-over of a device buffer of doubles to the host (no stream wait), see dv_publish."""
        self._check(self._lib.dv_publish(self._ctx, ctypes.c_void_p(int(device_ptr)), int(n_doubles)), "dv_publish")

    def publish_wait(self, out):
        """Poll for the last publish and copy it into `out` (float64, C-contiguous)."""
        self._check(self._lib.dv_publish_wait(self._ctx, N.f64ptr(out), out.size), "dv_publish_wait")
        return out

    def set_mailbox(self, address, n_bytes, rank, world):
        """Attach (address = 0: detach) the node's shared host segment of the mailbox exchange, see dv_set_mailbox."""
        self._check(self._lib.dv_set_mailbox(self._ctx, ctypes.c_void_p(int(address) or None), int(n_bytes), int(rank), int(world)),
                    "dv_set_mailbox")

    def mailbox_post(self, slot, seq):
        self._check(self._lib.dv_mailbox_post(self._ctx, int(slot), int(seq)), "dv_mailbox_post")

    def mailbox_wait(self, slot, seq, rank_mask, records, timeout_ms=20000):
        self._check(self._lib.dv_mailbox_wait(self._ctx, int(slot), int(seq), int(rank_mask), N.f64ptr(records),
                                              records.shape[1], int(timeout_ms)), "dv_mailbox_wait")
        return records

    def workgroup_shape(self, n_headings):
        """Shape of the scoring kernel in use for this many headings (0: not timed yet / not applicable)."""
        v = ctypes.c_int(0)
        self._check(self._lib.dv_workgroup_shape(self._ctx, int(n_headings), ctypes.byref(v)), "dv_workgroup_shape")
        return int(v.value)

    # -- error / coverage metrics on the device (NavBySceneFamiliarity.py:252-276) ----------------
    def set_training_path(self, points):
        """points: float64[n, 2] (x, y), or None to detach.  Clears the coverage marks."""
        if points is None:
            self._check(self._lib.dv_set_training_path(self._ctx, None, 0), "dv_set_training_path")
            return
        pts = np.ascontiguousarray(points, dtype=np.float64)
        assert pts.ndim == 2 and pts.shape[1] == 2
        self._check(self._lib.dv_set_training_path(self._ctx, N.f64ptr(pts), pts.shape[0]), "dv_set_training_path")

    def path_error_enqueue(self, x, y, reach):
        rc = self._lib.dv_path_error_enqueue(self._ctx, x, y, reach)
        if rc:
            self._check(rc, "dv_path_error_enqueue")

    def path_error_wait(self):
        out = self._err_out
        rc = self._lib.dv_path_error_wait(self._ctx, self._err_out_ref)
        if rc:
            self._check(rc, "dv_path_error_wait")
        return out.value

    def path_coverage(self, n):
        out = np.empty(int(n), dtype=np.uint8)
        self._check(self._lib.dv_path_coverage(self._ctx, N.u8ptr(out), int(n)), "dv_path_coverage")
        return out.astype(bool)

    def patches_on_level(self):
        """True when the resident patches allowed the fp4 form of the matrix-core kernel (dv_patches_on_level)."""
        rc = self._lib.dv_patches_on_level(self._ctx)
        if rc < 0:
            self._check(rc, "dv_patches_on_level")
        return bool(rc)

    def scoring_form(self):
        """Form of the last integer scoring pass (dv_scoring_form): matrix cores / fp4 coefficients / fused finishing."""
        rc = self._lib.dv_scoring_form(self._ctx)
        if rc < 0:
            self._check(rc, "dv_scoring_form")
        return dict(matrix_cores=bool(rc & 1), fp4=bool(rc & 2), fused_finish=bool(rc & 4))

    def path_reset(self):
        self._check(self._lib.dv_path_reset(self._ctx), "dv_path_reset")

    # update_error for the agents of an ensemble: a coverage array per agent (slot) on the device
    def path_slots(self, n_slots):
        self._check(self._lib.dv_path_slots(self._ctx, int(n_slots)), "dv_path_slots")

    def path_error_batch(self, slots, xs, ys, reach):
        """nearest[i] of agent slots[i] at (xs[i], ys[i]) (NavBySceneFamiliarity.py:252-276 for all of them at once); its marks updated."""
        slots = np.ascontiguousarray(slots, dtype=np.int32)
        xs = np.ascontiguousarray(xs, dtype=np.float64)
        ys = np.ascontiguousarray(ys, dtype=np.float64)
        out = np.empty(len(slots), dtype=np.float64)
        self._check(self._lib.dv_path_error_batch(self._ctx, slots.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)), N.f64ptr(xs), N.f64ptr(ys),
                                                  len(slots), float(reach), N.f64ptr(out)), "dv_path_error_batch")
        return out

    def path_coverage_slot(self, slot, n):
        out = np.empty(int(n), dtype=np.uint8)
        self._check(self._lib.dv_path_coverage_slot(self._ctx, int(slot), N.u8ptr(out), int(n)), "dv_path_coverage_slot")
        return out.astype(bool)

    def path_reset_slot(self, slot=-1):
        self._check(self._lib.dv_path_reset_slot(self._ctx, int(slot)), "dv_path_reset_slot")

    def stream_read_gbps(self, n_bytes=1 << 30, iters=10):
        g = ctypes.c_double(0)
        self._check(self._lib.dv_stream_read_gbps(self._ctx, int(n_bytes), int(iters), ctypes.byref(g)),
                    "dv_stream_read_gbps")
        return float(g.value)
