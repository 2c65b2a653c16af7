"""navsim.util counterpart: the familiarity-model plug-in, backed by the HIP engine.

Reference: navsim/util.pyx:10-25 (`sads_familiarity(chem_weight)` -> `internal(scenes)` -> `func`).
"""
import numpy as np

from .engine import FamiliarityEngine


def sads_familiarity(chem_weight=0.0, device=0, exact=False, devices=None):
    """Two-stage factory with the reference's shape.

    stage 1  sads_familiarity(chem_weight)          binds the weight          (util.pyx:10)
    stage 2  model(scenes: uint8[F,h,w,3]) -> func  uploads the library once  (util.pyx:11-13)
    func(scene: uint8[h,w,3], fambuf: float64[F])   writes fambuf in place    (util.pyx:14-20)
    func.max_familiarity = h*w                                                (util.pyx:22)

    Extras carried by `func` (used by the agent's fused step): func.engine, func.chem_weight.
    `exact=True` makes every fambuf value the reference's double bit for bit (slower fp64 kernel);
    the default integer-sum scores are within 1e-12 relative of it.
    `devices=[d0, d1, ...]`: ONE process, the library cut into contiguous blocks over these devices
    (navsim_amd.group.FamiliarityGroup over a dv_group); func.engine.step is then the merged, unsharded decision.  The agent
    senses its patches with the host sensor model in that form.
    """
    if devices is not None:
        return _group_sads_familiarity(chem_weight, list(devices))

    def bind(engine, scenes):
        maxfam = scenes[0].shape[0] * scenes[0].shape[1]

        def func(scene, fambuf):
            engine.score(scene, fambuf)

        func.max_familiarity = maxfam
        func.engine = engine
        func.chem_weight = chem_weight
        return func

    def sads_familiarity_internal(scenes):
        assert 0 <= chem_weight <= 1
        engine = FamiliarityEngine(device=device, exact=exact)
        engine.set_library(scenes, chem_weight)
        return bind(engine, scenes)

    # Hooks for navsim_amd.NavBySceneFamiliarity: it creates the engine early (landscape and sensor model live on
    # the GPU too), builds the library on the device and then binds `func` to that engine.
    def make_engine():
        assert 0 <= chem_weight <= 1
        return FamiliarityEngine(device=device, exact=exact)

    sads_familiarity_internal.make_engine = make_engine
    sads_familiarity_internal.from_engine = bind
    sads_familiarity_internal.chem_weight = chem_weight
    return sads_familiarity_internal


def _group_sads_familiarity(chem_weight, devices):
    def sads_familiarity_internal(scenes):
        assert 0 <= chem_weight <= 1
        from .group import FamiliarityGroup
        group = FamiliarityGroup(devices)
        group.set_library(scenes, chem_weight)

        def func(scene, fambuf):
            group.score(scene, fambuf)

        func.max_familiarity = scenes[0].shape[0] * scenes[0].shape[1]
        func.engine = group
        func.chem_weight = chem_weight
        return func

    sads_familiarity_internal.chem_weight = chem_weight
    return sads_familiarity_internal


hip_sads_familiarity = sads_familiarity


def ssd_familiarity(channel=2, device=0):
    """The north star's literal metric -- the pixel-wise sum of squared differences, the reference's `ssds`
    (navsim/util.pyx:171-184) -- as a familiarity plug-in of the reference's shape (util.pyx:10-25):

    stage 1  ssd_familiarity(channel)                  picks the compared channel of HSV scenes (0 H, 1 S, 2 V)
    stage 2  model(scenes) -> func                     uploads the library once:
                 uint8[F,h,w,3] / uint8[F,h,w]         -> the ssd_u8 metric: exact integer sums on the int8 matrix cores
                 float32[F,h,w]                        -> the ssd_f32 metric (within 1e-6 relative of ssds on the upcast data)
    func(scene, fambuf: float64[F])                    writes fambuf[f] = -SSD(scene, view f) in place: the most familiar view
                                                       is the one with the least SSD, as np.max / np.argmax of the agent's loop
                                                       (NavBySceneFamiliarity.py:313,315) expect
    func.max_familiarity = 0.0                         (an identical scene)

    Extras carried by `func` for the agent's fused step: func.engine, func.metric ("ssd_u8" / "ssd_f32"), func.channel.
    """
    if channel not in (0, 1, 2):
        raise ValueError("channel must be 0 (H), 1 (S) or 2 (V), got %r" % (channel,))

    def plane(a, lead):
        """The compared plane of a scene array: [.., h, w, 3] -> [.., h, w]; a single-channel array as it is."""
        a = np.asarray(a)
        if a.ndim == lead + 3:
            a = a[..., channel]
        if a.ndim != lead + 2:
            raise ValueError("scene array has shape %r" % (a.shape,))
        return np.ascontiguousarray(a)

    def bind(engine, metric):
        def func(scene, fambuf):
            if not (isinstance(fambuf, np.ndarray) and fambuf.dtype == np.float64):
                raise ValueError("Buffer dtype mismatch for fambuf, expected 'double'")
            p = plane(scene, 0)
            if metric == "ssd_u8":
                engine.score_u8(p, fambuf)
            else:
                engine.score_f32(p, fambuf)
            np.negative(fambuf, out=fambuf)

        func.max_familiarity = 0.0
        func.engine = engine
        func.metric = metric
        func.channel = channel
        return func

    def ssd_familiarity_internal(scenes):
        scenes = np.asarray(scenes)
        engine = FamiliarityEngine(device=device)
        if scenes.dtype == np.uint8:
            engine.set_library_u8(plane(scenes, 1))
            return bind(engine, "ssd_u8")
        if scenes.dtype == np.float32:
            engine.set_library_f32(plane(scenes, 1))
            return bind(engine, "ssd_f32")
        engine.close()
        raise ValueError("Buffer dtype mismatch, expected 'uint8_t' or 'float' but got '%s'" % scenes.dtype)

    # hooks for navsim_amd.NavBySceneFamiliarity (landscape, sensor model and library on the GPU: see sads_familiarity)
    ssd_familiarity_internal.make_engine = lambda: FamiliarityEngine(device=device)
    ssd_familiarity_internal.from_engine = lambda engine, scenes: bind(engine, "ssd_u8")
    ssd_familiarity_internal.metric = "ssd"
    ssd_familiarity_internal.channel = channel
    return ssd_familiarity_internal
