"""NavBySceneFamiliarity -- the navsim agent API on top of the MI355X familiarity engine.

Keeps the reference agent's constructor arguments, attributes, methods and stop exceptions
(navsim/NavBySceneFamiliarity.py:22-329) so that experiment drivers written for navsim run
unchanged; the heading loop of `step_forward` (:283-316) becomes ONE call into the HIP engine.
Plotting/animation (:332-661) is out of scope.

The sensor model (`get_sensor_mat`, :151-192, with util.pyx:91-168 behind it) runs on the GPU next to the
scoring kernel when the HIP model is used (landscape resident in HBM, `k_sense`), and is also restated here in
NumPy for other plug-ins; both are pinned byte-for-byte by tests/golden/t5_sensor.npz.
"""
import math
import os

import numpy as np

from .util import sads_familiarity


# ---- stop conditions (names and codes of navsim/NavBySceneFamiliarity.py:22-49) ----------------
class StopNavigationException(Exception):
    def get_reason(self):
        raise NotImplementedError()

    def get_code(self):
        raise NotImplementedError()

    def __str__(self):
        return self.get_reason()


class ReachedEndOfTrainingPathException(StopNavigationException):
    def get_reason(self):
        return "agent reached end of training path"

    def get_code(self):
        return 1


class NavigatingFailedException(StopNavigationException):
    pass


class TooFarFromTrainingPathException(NavigatingFailedException):
    def get_reason(self):
        return "agent went too far from training path"

    def get_code(self):
        return -1


class OutOfLandscapeBoundsException(NavigatingFailedException):
    def get_reason(self):
        return "agent went too close to boundary of landscape"

    def get_code(self):
        return -2


# ---- sensor model ------------------------------------------------------------------------------
def _c_round(v):
    """C round(): nearest integer, halves away from zero (util.pyx:122,131,166-167)."""
    t = np.trunc(v)
    return t + np.where(np.abs(v - t) >= 0.5, np.copysign(1.0, v), 0.0)


def fill_sensor_from(sensor, xpos, ypos, angle, landscape):
    """Rotated nearest-neighbour crop of the landscape into `sensor` (util.pyx:137-168).

    Sensor pixel (i, j) looks at offset (j - w/2, i - h/2) rotated by -(pi/2 - angle); the
    landscape is indexed [y, x].  Like the reference (bounds check on, wraparound on) negative
    indices wrap silently and indices past the end raise IndexError.
    """
    rot = -(0.5 * math.pi - angle)
    c, s = math.cos(rot), math.sin(rot)          # libm, as the reference's cimported cos/sin
    n0, n1 = sensor.shape[0], sensor.shape[1]
    px = np.arange(n1, dtype=np.float64)[None, :] - 0.5 * n1
    py = np.arange(n0, dtype=np.float64)[:, None] - 0.5 * n0
    rx = px * c - py * s
    ry = px * s + py * c
    ix = _c_round(rx + xpos).astype(np.int64)
    iy = _c_round(ry + ypos).astype(np.int64)
    sensor[...] = landscape[iy, ix]


def downscale_chem(image, factor_rows, factor_cols):
    """Block downscale with chemistry semantics (util.pyx:91-134).

    Per block: V = round(mean V); H = hue with the largest summed saturation (first maximum, hue 0
    on all-zero); S = round((sum // factor_rows) * factor_cols) wrapped to uint8 -- the
    reference's `conc / factor_rows*factor_cols` is C integer division followed by a multiply
    (util.pyx:131), and its uint8 cast wraps modulo 256.
    """
    fr, fc = int(factor_rows), int(factor_cols)
    nb_r, nb_c = image.shape[0] // fr, image.shape[1] // fc
    out = np.empty((nb_r, nb_c, image.shape[2]), dtype=np.uint8)
    blocks = image[:nb_r * fr, :nb_c * fc].reshape(nb_r, fr, nb_c, fc, image.shape[2])
    blocks = blocks.transpose(0, 2, 1, 3, 4).reshape(nb_r, nb_c, fr * fc, image.shape[2])
    vsum = blocks[..., 2].astype(np.float64).sum(axis=2)          # integer-valued: exact in any order
    out[..., 2] = _c_round(vsum / (fr * fc)).astype(np.int64).astype(np.uint8)
    if fr * fc == 1:
        out[..., 0] = blocks[..., 0, 0]
        # hue of a zero-saturation pixel loses the argmax to hue 0 (util.pyx:126-130)
        out[..., 0][blocks[..., 0, 1] == 0] = 0
        out[..., 1] = blocks[..., 0, 1]
        return out
    conc = np.zeros((nb_r, nb_c, 256), dtype=np.int64)
    rr, cc = np.meshgrid(np.arange(nb_r), np.arange(nb_c), indexing="ij")
    for k in range(fr * fc):
        np.add.at(conc, (rr, cc, blocks[..., k, 0].astype(np.int64)), blocks[..., k, 1].astype(np.int64))
    which = np.argmax(conc, axis=2)
    top = np.take_along_axis(conc, which[..., None], axis=2)[..., 0]
    out[..., 0] = which.astype(np.uint8)
    out[..., 1] = ((top // fr) * fc & 0xFF).astype(np.uint8)
    return out


class NavBySceneFamiliarity(object):
    """Agent that walks a landscape by scene familiarity (navsim/NavBySceneFamiliarity.py:57-329).

    `familiarity_model` is the reference's plug-in point: any `model(scenes) -> func(scene, fambuf)`
    with `func.max_familiarity` works (:72,140,299).  The default is the HIP engine; when the model's
    `func` carries an `.engine`, `step_forward` scores all headings with one fused device step.
    `track_scene_familiarity=False` skips the per-view minimum the reference only plots (:301-303); with True (the reference's
    default) the lean device step is kept and the minimum of the last step is worked out when `scene_familiarity` is read.
    """

    def __init__(self,
                 landscape,
                 sensor_dimensions,
                 step_size,
                 n_test_angles=60,
                 sensor_pixel_dimensions=[1, 1],
                 max_distance_to_training_path=np.inf,
                 n_sensor_levels=5,
                 mask_middle_n=0,
                 threshold_factor=2.,
                 coverage_threshold_factor=0.8,
                 saccade_degrees=180.,
                 sensor_px_per_mm=None,
                 familiarity_model=None,
                 track_scene_familiarity=True,
                 use_gpu_sensor=True):
        self.landscape = landscape
        self.position = (0., 0.)
        self.angle = 0.

        self.n_test_angles = n_test_angles
        self.mask_middle_n = mask_middle_n
        self.threshold_factor = threshold_factor
        self.coverage_threshold_factor = coverage_threshold_factor
        self.sensor_px_per_mm = sensor_px_per_mm
        self.track_scene_familiarity = track_scene_familiarity

        self.saccade_degrees = saccade_degrees
        half_sweep = saccade_degrees / 2
        self.angle_offsets = np.linspace(-(np.pi * half_sweep / 180.), np.pi * half_sweep / 180., self.n_test_angles)

        self.sensor_dimensions = np.asarray(sensor_dimensions)
        self.sensor_pixel_dimensions = np.asarray(sensor_pixel_dimensions)
        extent = self.sensor_dimensions * self.sensor_pixel_dimensions      # landscape px, (w, h)
        assert np.all(extent % 2 == 0)
        self._sensor_r = np.max(extent / 2)
        self._bounds = None
        self._bounds_arr = None
        self._roundbuf = np.empty((self.sensor_dimensions[1], self.sensor_dimensions[0]), dtype=np.float32)
        self._end_buf = np.empty(2, dtype=np.float64)
        self._spec = None                      # the pose and offsets the engine was asked to begin the next step for (_move)
        self._metric_slot = None               # member of an ensemble: its coverage marks' slot on the device (NavEnsemble)
        self._ens = None
        self.pipeline_steps = os.environ.get("DEJAVU_AGENT_PIPELINE", "1") != "0"
        self.lazy_scene = os.environ.get("DEJAVU_LAZY_SCENE", "1") != "0"      # track_scene_familiarity=True: the minimum is worked out when read
        self._landscape_glimpse_buf = np.empty((extent[1], extent[0], 3), dtype=np.uint8)
        self.n_sensor_pixels = np.prod(self.sensor_dimensions)

        if not isinstance(n_sensor_levels, tuple):
            n_sensor_levels = (256, 256, n_sensor_levels)
        assert len(n_sensor_levels) == 3
        assert all(2 <= l <= 256 for l in n_sensor_levels)
        self.n_sensor_levels = n_sensor_levels

        self.step_size = step_size
        self.angle_familiarity = np.empty(n_test_angles)
        self.step_familiarity = np.inf
        self.max_distance_to_training_path = max_distance_to_training_path

        self.clear_training()
        self.familiarity_model = familiarity_model if familiarity_model is not None else sads_familiarity()
        self.reset_error()

        # With the HIP model the landscape and the sensor model live on the GPU as well: patches are sensed where
        # they are scored.  Any other plug-in keeps the host sensor model below (same bytes, pinned by the fixtures).
        self._engine = None
        self.use_gpu_sensor = bool(use_gpu_sensor)
        make_engine = getattr(self.familiarity_model, "make_engine", None)
        if make_engine is not None and use_gpu_sensor and self.landscape.dtype == np.uint8:
            self._engine = make_engine()
            self._engine.set_landscape(self.landscape)
            self._engine.configure_sensor(self.sensor_dimensions, self.sensor_pixel_dimensions,
                                          self._level_tables(), self.mask_middle_n)

    def _level_tables(self):
        """uint8[3][256]: the float32 level quantisation of get_sensor_mat (:176-186) applied to every byte value."""
        lut = np.empty((3, 256), dtype=np.uint8)
        for ch in range(3):
            levels = self.n_sensor_levels[ch]
            buf = np.arange(256, dtype=np.uint8).astype(np.float32)
            buf /= 255
            buf *= (levels - 1)
            np.rint(buf, out=buf)
            buf /= (levels - 1)
            buf *= 255
            lut[ch] = buf          # truncating float32 -> uint8 cast
        return lut

    def _bounds_tuple(self):
        b = self._bounds
        if b is None:                          # the four limits of :153-158 as Python floats (same comparisons, no NumPy scalars per step)
            r = self._sensor_r
            ldims = self.landscape.shape
            b = self._bounds = (float(r), float(ldims[1] - r), float(ldims[0] - r))
            self._bounds_arr = np.array(b, dtype=np.float64)
        return b

    def _check_bounds(self, position):
        b = self._bounds_tuple()
        if (position[0] <= b[0]) or (position[1] <= b[0]) or (position[0] >= b[1]) or (position[1] >= b[2]):
            raise OutOfLandscapeBoundsException()

    # ---- training (:118-148) -------------------------------------------------------------------
    def train_from_path(self, points):
        if self.training_path is not None:
            raise ValueError("Tried to train NavBySceneFamiliarity more than once.")
        points = np.asarray(points)
        n = len(points)
        self.familiar_scenes = np.empty((n, self.sensor_dimensions[1], self.sensor_dimensions[0],
                                         self.landscape.shape[2]), dtype=self.landscape.dtype)
        steps = points[1:] - points[:-1]
        self.training_path_length = np.sum(np.linalg.norm(steps, axis=1))
        headings = np.arctan2(steps[:, 1], steps[:, 0])
        # view i looks towards point i+1; the last point reuses the last heading (:129-132)
        view_headings = headings[np.minimum(np.arange(n), n - 2)]
        if self._engine is not None:
            for pt in points:
                self._check_bounds(pt)
            # sensed and ingested on the device; familiar_scenes comes back for the API
            if getattr(self.familiarity_model, "metric", "sads_hsv") == "ssd":
                self.familiar_scenes[...] = self._engine.set_library_u8_from_poses(
                    points[:, 0], points[:, 1], view_headings, self.familiarity_model.channel)
            else:
                self.familiar_scenes[...] = self._engine.set_library_from_poses(
                    points[:, 0], points[:, 1], view_headings, self.familiarity_model.chem_weight)
        else:
            for i in range(n):
                self.familiar_scenes[i] = self.get_sensor_mat(points[i], view_headings[i])

        self.scene_familiarity = np.zeros(n, dtype=np.float64)
        self._scene_is_inf = False
        self.training_path = points
        self._metrics_on_device = False
        if self._engine is not None and hasattr(self._engine, "set_training_path") and points.shape[1] == 2:
            self._engine.set_training_path(points)
            self._metrics_on_device = True
        self.reset_error()
        # library hand-off (:140): already resident when the views were sensed on the GPU
        if self._engine is not None:
            self._familiarity_func = self.familiarity_model.from_engine(self._engine, self.familiar_scenes)
        else:
            self._familiarity_func = self.familiarity_model(self.familiar_scenes)
            # a plug-in over several devices (util.sads_familiarity(cw, devices=[...]) -> group.FamiliarityGroup) takes a copy of the
            # landscape and the sensor's tables on every member, so that a step senses on the devices: nothing but the pose goes up
            eng = getattr(self._familiarity_func, "engine", None)
            if self.use_gpu_sensor and hasattr(eng, "attach_sensor") and self.landscape.dtype == np.uint8:
                eng.attach_sensor(self.landscape, self.sensor_dimensions, self.sensor_pixel_dimensions, self._level_tables(), self.mask_middle_n)

    def train_additional_path(self, points):
        """A further training path behind the first (the reference's experiment script anticipates several per
        library, scripts/run_experiment.py:23-26, but its class has no call for it): the new points' views are appended
        to the library -- on the device only the new view groups are re-tiled -- and the path, its length and the
        coverage marks grow with them.  Views are taken as in train_from_path (:124-132): each point looks towards the
        next one of ITS path, the last reuses the last heading."""
        if self.training_path is None:
            raise ValueError("train_from_path first")
        points = np.asarray(points, dtype=np.float64)
        n = len(points)
        steps = points[1:] - points[:-1]
        headings = np.arctan2(steps[:, 1], steps[:, 0])
        view_headings = headings[np.minimum(np.arange(n), n - 2)]
        engine = getattr(self._familiarity_func, "engine", None)
        ssd = str(getattr(self._familiarity_func, "metric", "")).startswith("ssd")
        if self._engine is not None and engine is self._engine and not ssd:
            for pt in points:
                self._check_bounds(pt)
            new_views = self._engine.append_library_from_poses(points[:, 0], points[:, 1], view_headings)
        elif ssd:
            # the SSD libraries have no append: sense the new views and ingest everything again
            new_views = np.stack([self.get_sensor_mat(points[i], view_headings[i]) for i in range(n)])
            both = np.concatenate([self.familiar_scenes, new_views])
            if self._familiarity_func.metric == "ssd_u8":
                engine.set_library_u8(np.ascontiguousarray(both[..., self._familiarity_func.channel]))
            else:
                engine.set_library_f32(np.ascontiguousarray(both[..., self._familiarity_func.channel]
                                                            if both.ndim == 4 else both))
        else:
            new_views = np.stack([self.get_sensor_mat(points[i], view_headings[i]) for i in range(n)])
            if engine is not None and hasattr(engine, "append_library"):
                engine.append_library(new_views)
        self.familiar_scenes = np.concatenate([self.familiar_scenes, new_views])
        if engine is None or (not ssd and not hasattr(engine, "append_library")):
            self._familiarity_func = self.familiarity_model(self.familiar_scenes)     # any other plug-in: hand it all views again
        self.training_path_length = self.training_path_length + np.sum(np.linalg.norm(steps, axis=1))
        self.training_path = np.concatenate([self.training_path, points])
        self.scene_familiarity = np.zeros(len(self.training_path), dtype=np.float64)
        self._scene_is_inf = False
        if self._metrics_on_device:
            self._engine.set_training_path(self.training_path)
        self.reset_error()

    def clear_training(self):
        func = getattr(self, "_familiarity_func", None)
        if func is not None and hasattr(func, "engine"):
            if func.engine is getattr(self, "_engine", None):
                func.engine.clear_library()          # keep the landscape and the sensor configuration
            else:
                func.engine.close()
        if getattr(self, "_metrics_on_device", False) and getattr(self, "_engine", None) is not None:
            self._engine.set_training_path(None)
        self._metrics_on_device = False
        self.training_path = None
        self.familiar_scenes = None
        self._familiarity_func = None
        self.scene_familiarity = None
        self._scene_is_inf = False
        self.training_path_length = None

    # ---- sensor (:151-192) ---------------------------------------------------------------------
    def get_sensor_mat(self, position, angle):
        self._check_bounds(position)
        if self._engine is not None:
            return self._engine.sense([position[0]], [position[1]], [angle])[0]

        fill_sensor_from(self._landscape_glimpse_buf, position[0], position[1], angle, self.landscape)
        out = downscale_chem(self._landscape_glimpse_buf,
                             self.sensor_pixel_dimensions[1], self.sensor_pixel_dimensions[0])

        # quantise every channel to its number of levels, through float32 like the reference (:176-186)
        buf = self._roundbuf
        for ch in range(3):
            levels = self.n_sensor_levels[ch]
            buf[:] = out[:, :, ch]
            buf /= 255
            buf *= (levels - 1)
            np.rint(buf, out=buf)
            buf /= (levels - 1)
            buf *= 255
            out[:, :, ch] = buf          # truncating float32 -> uint8 cast

        mid = out.shape[1] // 2
        out[:, mid - self.mask_middle_n:mid + self.mask_middle_n] = 0
        return out

    # ---- error / coverage metrics (:195-276) ---------------------------------------------------
    def reset_error(self):
        self.stopped_with_exception = None
        self.navigated_for_frames = 0
        self._navigation_error = 0.0
        self._n_navigation_error = 0
        self._pending_errors = 0
        self._err_positions = []               # positions whose metrics are outstanding, in the order asked
        self._last_nearest = None              # (distance, position) of the last answer (the bound of _can_defer_error)
        self._error_pos = None                 # a position whose metrics are still to be asked for (the lean step defers them by one call)
        self._host_coverage = None
        if self.training_path is not None:
            self._host_coverage = np.zeros(len(self.training_path), dtype=bool)
            if getattr(self, "_metric_slot", None) is not None:
                self._ens._drop_errors(self)
                self._engine.path_reset_slot(self._metric_slot)
            elif getattr(self, "_metrics_on_device", False):
                self._engine.path_reset()

    # The metrics of update_error (:252-276) run on the device when the agent owns its engine: a step asks for them
    # (dv_path_error_enqueue) and collects the answer one step later, or when a metric is read -- except with a finite
    # max_distance_to_training_path, where the reference may stop the run inside update_error and the answer is
    # awaited at once.  Every number is the reference's double arithmetic; only the moment it reaches the host moves.
    def _flush_error_pos(self):
        if self._error_pos is not None:
            self._engine.path_error_enqueue(self._error_pos[0], self._error_pos[1], self.coverage_threshold_factor * self.step_size)
            self._error_asked(self._error_pos)
            self._error_pos = None

    def _take_error(self, nearest):
        if nearest > self.max_distance_to_training_path:
            raise TooFarFromTrainingPathException()
        self._navigation_error += nearest * nearest
        self._n_navigation_error += 1

    def _error_asked(self, pos):
        """A position's metrics were asked of the device; the answers come back in the order asked."""
        self._pending_errors += 1
        self._err_positions.append((float(pos[0]), float(pos[1])))

    def _error_answer(self, nearest):
        self._pending_errors -= 1
        pos = self._err_positions.pop(0) if self._err_positions else None
        self._take_error(nearest)
        self._last_nearest = (nearest, pos)

    def _can_defer_error(self, position):
        """May the metrics of the position the coming step ends at be collected one step late?  Always with an infinite
        max_distance_to_training_path.  With a finite one the reference stops the run inside update_error (:264), in the very step that
        gets too far -- so only while that cannot happen: the new position is at most step_size from `position`, and a point's
        distance to the path is at most the last known one plus the way from where that was measured (triangle inequality)."""
        if not self._metrics_on_device:
            return False
        m = self.max_distance_to_training_path
        if not math.isfinite(m):
            return True
        ln = self._last_nearest
        if ln is None or ln[1] is None:
            return False
        return ln[0] + math.hypot(position[0] - ln[1][0], position[1] - ln[1][1]) + self.step_size <= m * (1.0 - 1e-9) - 1e-9

    def _collect_errors(self, keep=0):
        self._flush_error_pos()
        while self._pending_errors > keep:
            self._error_answer(self._engine.path_error_wait())

    @property
    def _coverage_array(self):
        if getattr(self, "_metric_slot", None) is not None and self.training_path is not None:
            self._ens._flush_errors()
            return self._engine.path_coverage_slot(self._metric_slot, len(self.training_path))
        if getattr(self, "_metrics_on_device", False) and self.training_path is not None:
            self._collect_errors()
            return self._engine.path_coverage(len(self.training_path))
        return self._host_coverage

    @property
    def navigation_error(self):
        if getattr(self, "_metric_slot", None) is not None:
            self._ens._flush_errors()
        elif getattr(self, "_metrics_on_device", False):
            self._collect_errors()
        return np.sqrt(self._navigation_error / self._n_navigation_error)

    @property
    def percent_recapitulated(self):
        cov = self._coverage_array
        return np.sum(cov) / len(cov)

    def _window(self, n_consecutive_scenes):
        return int(n_consecutive_scenes * len(self.training_path))

    def percent_recapitulated_forgiving(self, n_consecutive_scenes=0.05):
        """Furthest point i/F of the path such that the `window` scenes before i are all covered (:218-232)."""
        win = self._window(n_consecutive_scenes)
        total = len(self.training_path)
        cov = self._coverage_array
        for i in range(total, win - 1, -1):
            if np.all(cov[i - win:i]):
                return i / total
        return 0.

    def n_captures(self, n_consecutive_scenes=0.05):
        """Times the agent got onto the path after being off it (:235-249)."""
        win = self._window(n_consecutive_scenes)
        cov = self._coverage_array
        count = 0
        for i in range(len(self.training_path) - win):
            if (not cov[i]) and np.all(cov[i + 1:i + 1 + win]):
                count += 1
        return count

    def update_error(self):
        self.navigated_for_frames += 1
        if getattr(self, "_metric_slot", None) is not None:
            # a member of an ensemble: its position goes on the ensemble's list, and the metrics of all members are taken in one
            # device call at the end of the ensemble's step (NavEnsemble._flush_errors) -- at once when the agent steps on its own
            self._ens._want_error(self)
            if not self._ens._stepping:
                self._ens._flush_errors(raise_for=self)
            return
        if getattr(self, "_metrics_on_device", False):
            self._collect_errors()                               # the previous step's answer: ready by now (and a deferred position first)
            self._engine.path_error_enqueue(self.position[0], self.position[1],
                                            self.coverage_threshold_factor * self.step_size)
            self._error_asked(self.position)
            if math.isfinite(self.max_distance_to_training_path):
                self._collect_errors()                           # may raise TooFarFromTrainingPathException here (:264)
            return
        delta = self.training_path - self.position
        delta *= delta
        dist = np.sqrt(np.sum(delta, axis=1))
        nearest = np.min(dist)
        if nearest > self.max_distance_to_training_path:
            raise TooFarFromTrainingPathException()
        self._navigation_error += nearest * nearest
        self._n_navigation_error += 1
        reach = self.coverage_threshold_factor * self.step_size
        if nearest <= reach:
            self._host_coverage |= (dist <= reach)

    # ---- scene_familiarity: kept every step, or worked out when it is read ----------------------
    _scene_fam = None
    _scene_stale = None

    @property
    def scene_familiarity(self):
        """float64[F]: min over the headings of the last step's per-view scores (:287,301-303).  With the lean device step
        (`lazy_scene`, the default) a step only notes its pose; the first read senses that pose again and takes the minimum then --
        the same patches, the same numbers, paid by whoever reads them."""
        st = self._scene_stale
        if st is not None:
            self._scene_stale = None
            res = self._engine.sense_step(st[0], st[1], (st[2] + self.angle_offsets) % (2 * np.pi), want_scene=True)
            self._scene_fam[:] = res["scene_familiarity"]
        return self._scene_fam

    @scene_familiarity.setter
    def scene_familiarity(self, value):
        self._scene_stale = None
        self._scene_fam = value

    # ---- the step (:279-329) -------------------------------------------------------------------
    def step_forward(self, fake=False):
        position = self.position
        self.angle_familiarity[:] = np.nan
        assert len(self.familiar_scenes) == len(self._scene_fam)

        func = self._familiarity_func
        engine = getattr(func, "engine", None)
        defer_error = False
        begin_next = False
        cand = None
        scene_stale = None
        self._scene_stale = None               # (whatever was left to work out belonged to the step before)
        begun_next = None                      # (the next step was begun by the call that ended this one, an error answer it collected)
        if engine is not None and str(getattr(func, "metric", "")).startswith("ssd"):
            self._step_ssd(func, engine, position)
            best_idex = self.last_scored_idex
        elif engine is not None:
            # one fused device step for all headings: kernel + min-merge + max + argmax (:289-315)
            try:
                if engine is self._engine:
                    # patches are sensed on the GPU, straight into the scoring kernel's operand layout
                    self._check_bounds(position)
                    lazy = self.track_scene_familiarity and self.lazy_scene
                    if (lazy or not self.track_scene_familiarity) and hasattr(engine, "agent_step") and self.n_test_angles <= 64:
                        # the lean form of the same device step (dv_agent_step): ONE call does the sensing, the scoring and the
                        # device side of the error metrics -- the answer asked for at the last step is collected, the position
                        # the last step ended at is handed in -- and writes the per-heading maxima straight into angle_familiarity
                        # The step may already be on the device: the last step began it as soon as this pose was known (_move), so
                        # that its book-keeping, the caller's loop and this call's preamble ran beside the device's work.  It counts
                        # only if it was begun for exactly this pose and these offsets and nothing else was asked of the engine since.
                        best_idex = None
                        cand = None
                        spec, self._spec = self._spec, None
                        if spec is not None and spec == (position[0], position[1], self.angle, self.angle_offsets.tobytes()):
                            # while the device works: where each candidate heading would take the agent (:317-321 for every heading
                            # at once -- NumPy's elementwise loops give an element what they give the scalar), so that the pose of
                            # the heading chosen is three look-ups away when the record arrives
                            cand_angle = (self.angle + self.angle_offsets) % (2 * np.pi)
                            cand = (cand_angle, position[0] + self.step_size * np.cos(cand_angle),
                                    position[1] + self.step_size * np.sin(cand_angle))
                            lean_error = self._can_defer_error(position)
                            if self.pipeline_steps and (fake or lean_error) and self._error_pos is None:
                                # ... and the next step begun by the same call that hands this one's record out
                                # (dv_agent_step_end_begin): nothing of this interpreter between the record and the launch
                                got = engine.agent_step_end_begin(cand, self._bounds_arr, not fake, self.coverage_threshold_factor * self.step_size)
                                if got is not None:
                                    best_idex, begun_next = got[0], (got[1], got[2])
                            else:
                                best_idex = engine.agent_step_end()
                            if best_idex is None:
                                cand = None
                        if best_idex is None:
                            epos = self._error_pos
                            best_idex, nearest = engine.agent_step(position[0], position[1], self.angle, self.angle_offsets,
                                                                   self.angle_familiarity, epos,
                                                                   self.coverage_threshold_factor * self.step_size)
                            if epos is not None:
                                self._error_pos = None
                                self._error_asked(epos)
                            if nearest is not None:
                                self._error_answer(nearest)
                        res = None
                        defer_error = self._can_defer_error(position)
                        begin_next = self.pipeline_steps
                        if lazy:
                            # the per-view minimum of THIS step (:301-303) is worked out when scene_familiarity is read: the
                            # reference only plots it (:540,630), and keeping it every step halves the agent's rate
                            scene_stale = (position[0], position[1], self.angle)
                    elif not self.track_scene_familiarity and hasattr(engine, "sense_step_into"):
                        # (the same device step through the engine's lean binding: no per-step record, views or dictionary)
                        best_idex = engine.sense_step_into(position[0], position[1], self.angle, self.angle_offsets,
                                                           self.angle_familiarity)
                        res = None
                    else:
                        res = engine.sense_step(position[0], position[1], (self.angle + self.angle_offsets) % (2 * np.pi),
                                            want_scene=self.track_scene_familiarity)
                elif getattr(engine, "sensor_attached", False) and self.n_test_angles <= 64:
                    self._check_bounds(position)
                    res = engine.sense_step(position[0], position[1], (self.angle + self.angle_offsets) % (2 * np.pi),
                                            want_scene=self.track_scene_familiarity)
                else:
                    patches = np.empty((self.n_test_angles,) + self.familiar_scenes.shape[1:], dtype=np.uint8)
                    for a_idex, angle_offset in enumerate(self.angle_offsets):
                        patches[a_idex] = self.get_sensor_mat(position, (self.angle + angle_offset) % (2 * np.pi))
                    res = engine.step(patches, want_scene=self.track_scene_familiarity)
            except Exception:
                # the reference resets scene_familiarity to +inf before it senses (:287); a step that stops here
                # leaves it so.  (A completed step overwrites every entry, so the fill is not paid per step.)
                self._scene_fam[:] = np.inf
                self._scene_is_inf = True
                raise
            if res is not None:
                self.angle_familiarity[:] = res["angle_familiarity"]
                best_idex = res["best_idex"]
            if scene_stale is not None:
                self._scene_stale = scene_stale
                self._scene_is_inf = False
            elif self.track_scene_familiarity:
                self._scene_fam[:] = res["scene_familiarity"]
                self._scene_is_inf = False
            elif not self._scene_is_inf:
                self._scene_fam[:] = np.inf              # not tracked: stays at the reference's reset value
                self._scene_is_inf = True
        else:
            # any other plug-in: the reference's loop, one model call per heading
            self._scene_fam[:] = np.inf
            self._scene_is_inf = False
            temp_fam = np.empty_like(self._scene_fam)
            for a_idex, angle_offset in enumerate(self.angle_offsets):
                smat = self.get_sensor_mat(position, (self.angle + angle_offset) % (2 * np.pi))
                temp_fam[:] = np.nan
                func(smat, temp_fam)
                np.minimum(self._scene_fam, temp_fam, out=self._scene_fam)
                self.angle_familiarity[a_idex] = np.max(temp_fam)
            best_idex = np.argmax(self.angle_familiarity)

        self._move(best_idex, fake, defer_error, begin_next, cand, begun_next)

    def _step_ssd(self, func, engine, position):
        """The heading loop (:289-315) with the SSD plug-in (util.ssd_familiarity): ONE device step -- sense, score on the matrix
        cores, decide -- when the sensor model runs on the GPU; familiarity = -SSD, so the reference's max / argmax hold."""
        angles = (self.angle + self.angle_offsets) % (2 * np.pi)
        try:
            if engine is self._engine:
                self._check_bounds(position)
                res = engine.sense_step_u8(position[0], position[1], angles, func.channel,
                                           want_scene=self.track_scene_familiarity)
            else:
                patches = np.stack([self.get_sensor_mat(position, a) for a in angles])
                planes = np.ascontiguousarray(patches[..., func.channel] if patches.ndim == 4 else patches)
                if func.metric == "ssd_u8":
                    res = engine.step_u8(planes, want_scene=self.track_scene_familiarity)
                else:
                    res = engine.step_f32(planes, want_scene=self.track_scene_familiarity)
        except Exception:
            self._scene_fam[:] = np.inf
            self._scene_is_inf = True
            raise
        np.negative(res["angle_ssd"], out=self.angle_familiarity)
        if self.track_scene_familiarity:
            np.negative(res["scene_ssd"], out=self._scene_fam)     # min over headings of -SSD = -(max over headings of SSD)
            self._scene_is_inf = False
        elif not self._scene_is_inf:
            self._scene_fam[:] = np.inf
            self._scene_is_inf = True
        self.last_scored_idex = res["best_idex"]

    def _move(self, best_idex, fake=False, defer_error=False, begin_next=False, cand=None, begun_next=None):
        """The part of a step after the heading is chosen (:316-329): turn, advance, book-keeping, stop conditions.
        `cand`: (angles, xs, ys) of every candidate heading, worked out beforehand with the same operations."""
        position = self.position
        self.step_familiarity = self.angle_familiarity[best_idex]
        if cand is not None:
            angle = cand[0][best_idex]
            self.position = (cand[1][best_idex], cand[2][best_idex])
        else:
            angle = (self.angle + self.angle_offsets[best_idex]) % (2 * np.pi)
            self.position = (position[0] + self.step_size * np.cos(angle),
                             position[1] + self.step_size * np.sin(angle))
        self.angle = angle
        self.last_best_idex = int(best_idex)

        begun = False
        if begun_next is not None and begun_next[0]:
            # the device already has the next step (dv_agent_step_end_begin took the pose from the same candidates): the book-keeping
            # of agent_step_begin below, nothing else
            if not fake:
                self.navigated_for_frames += 1
                self._error_asked(self.position)
            if begun_next[1] is not None:
                self._error_answer(begun_next[1])
            self._spec = (self.position[0], self.position[1], angle, self.angle_offsets.tobytes())
            begun = True
        elif begin_next and (fake or defer_error):
            # the lean step on the agent's own engine: the NEXT step's device work starts now, for the pose just computed (the
            # heading update is all that serialises two steps, :317-323) -- unless that pose is out of bounds, where the next call
            # stops before it senses (:153-158).  The position's error metrics (update_error, :324) ride in the same launch.
            b = self._bounds
            npos = self.position
            if b is not None and not ((npos[0] <= b[0]) or (npos[1] <= b[0]) or (npos[0] >= b[1]) or (npos[1] >= b[2])):
                epos = None
                if not fake:
                    self.navigated_for_frames += 1
                    self._flush_error_pos()
                    epos = npos
                nearest = self._engine.agent_step_begin(npos[0], npos[1], angle, self.angle_offsets, self.angle_familiarity, epos,
                                                        self.coverage_threshold_factor * self.step_size)
                if epos is not None:
                    self._error_asked(epos)
                if nearest is not None:
                    self._error_answer(nearest)
                self._spec = (npos[0], npos[1], angle, self.angle_offsets.tobytes())
                begun = True

        if not fake:
            if begun:
                pass                                              # (asked for with the step just begun)
            elif defer_error:
                # the lean step: this position's metrics are asked for by the NEXT step's device call (or when a metric is read)
                self.navigated_for_frames += 1
                self._flush_error_pos()
                self._error_pos = self.position
            else:
                self.update_error()
            # np.linalg.norm of the 2-vector (:325-326) without its wrapper: the same subtraction, dot product and square root,
            # in one small buffer kept for the agent's lifetime
            d, end = self._end_buf, self.training_path[-1]
            d[0] = end[0] - self.position[0]
            d[1] = end[1] - self.position[1]
            if math.sqrt(d.dot(d)) <= self.threshold_factor * self.step_size:
                raise ReachedEndOfTrainingPathException()

    # ---- ensembles: several agents on one engine and one library (navsim_amd/ensemble.py) --------------------
    def clone_for_ensemble(self):
        """Another agent on the SAME engine, landscape and trained library, with its own pose and book-keeping."""
        import copy
        if self._engine is None or self.training_path is None:
            raise ValueError("clone a trained agent whose sensor model runs on the GPU")
        other = copy.copy(self)                                   # shares engine, library views, training path
        other._metrics_on_device = False                          # the device holds ONE agent's coverage marks: clones keep
                                                                  # theirs on the host (same arithmetic)
        other.angle_familiarity = np.full_like(self.angle_familiarity, np.nan)
        other.scene_familiarity = None if self.track_scene_familiarity is False else np.zeros_like(self._scene_fam)
        other.position, other.angle = None, None
        other._spec = None
        other._metric_slot = None
        other.step_familiarity = None
        other.reset_error()
        return other

    def headings_to_test(self):
        """(x, y, absolute test angles) of the coming step, after the reference's bounds check (:153-158, :289-293)."""
        self.angle_familiarity[:] = np.nan
        self._check_bounds(self.position)
        return self.position[0], self.position[1], (self.angle + self.angle_offsets) % (2 * np.pi)

    def apply_step_result(self, res, fake=False):
        """Finish a step whose device work was done elsewhere (an ensemble pass): same state changes as step_forward."""
        self.apply_step_arrays(res["angle_familiarity"], res["best_idex"], fake)

    def apply_step_arrays(self, angle_familiarity, best_idex, fake=False, cand=None):
        """`cand`: (angles, xs, ys) of this agent's candidate headings, worked out for all members at once (NavEnsemble)."""
        self.angle_familiarity[:] = angle_familiarity
        self._move(best_idex, fake, cand=cand)
