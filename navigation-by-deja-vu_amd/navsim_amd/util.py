"""navsim.util counterpart: the familiarity-model plug-in, backed by the HIP engine.

Reference: navsim/util.pyx:10-25 (`sads_familiarity(chem_weight)` -> `internal(scenes)` -> `func`).
"""
from .engine import FamiliarityEngine


def sads_familiarity(chem_weight=0.0, device=0, exact=False):
    """Two-stage factory with the reference's shape.

    stage 1  sads_familiarity(chem_weight)          binds the weight          (util.pyx:10)
    stage 2  model(scenes: uint8[F,h,w,3]) -> func  uploads the library once  (util.pyx:11-13)
    func(scene: uint8[h,w,3], fambuf: float64[F])   writes fambuf in place    (util.pyx:14-20)
    func.max_familiarity = h*w                                                (util.pyx:22)

    Extras carried by `func` (used by the agent's fused step): func.engine, func.chem_weight.
    `exact=True` makes every fambuf value the reference's double bit for bit (slower fp64 kernel);
    the default integer-sum scores are within 1e-12 relative of it.
    """
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


hip_sads_familiarity = sads_familiarity
