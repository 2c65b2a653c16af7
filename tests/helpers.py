"""Shared input regeneration for the parity tests (inputs come from seeds, outputs from fixtures)."""
import hashlib

import numpy as np

from navsim_amd import synth


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def kernel_case_inputs(case):
    F, h, w, seed = case["F"], case["h"], case["w"], case["seed"]
    if case["kind"] == "levels":
        lib = synth.synth_views(seed, F, h, w)
        scene = synth.synth_patches(seed, 1, h, w)[0]
    else:
        lib = synth.random_hsv(seed, (F, h, w, 3))
        scene = synth.random_hsv(seed + 1, (h, w, 3))
        lib[..., 0] &= 0x03
        scene[..., 0] &= 0x03
    assert sha(lib) == case["lib_sha"] and sha(scene) == case["scene_sha"], "input regeneration drifted"
    return lib, scene


def step_case_inputs(case):
    F, h, w, A, seed, kind = case["F"], case["h"], case["w"], case["A"], case["seed"], case["kind"]
    if kind == "random":
        lib = synth.random_hsv(seed, (F, h, w, 3))
        lib[..., 0] &= 0x07
        patches = synth.random_hsv(seed + 1, (A, h, w, 3))
        patches[..., 0] &= 0x07
    else:
        lib = synth.synth_views(seed, F, h, w)
        patches = synth.synth_patches(seed, A, h, w)
        if kind == "near":
            patches[3] = synth.near_match_patch(lib[F // 3], seed)
            patches[7] = synth.near_match_patch(lib[F // 2], seed + 9)
        if kind == "dup":
            lib[:] = lib[0]
            patches[:] = lib[0]
            patches[:, 0, 0, 2] = 255 - lib[0, 0, 0, 2]
            lib[7, 1, 1, 2] ^= 0xFF
    assert sha(lib) == case["lib_sha"] and sha(patches) == case["patches_sha"], "input regeneration drifted"
    return lib, patches


# Forms of the scoring path every GPU engine test runs under.  The engine reads these variables when it is created.
#   k_finish / k_combine+k_tail : the two ways a step ends (DEJAVU_FINISH=2 / 0), byte-plane kernels timed as usual;
#   mfma    : the bit-plane copy is built whenever the library's values allow it (DEJAVU_BITS=2) and scored on the matrix
#             cores (DEJAVU_SHAPE=6, k_sad_mfma_dual); libraries it cannot describe fall back to the byte-plane kernels;
#             finishing epilogue outside the kernel (DEJAVU_FUSE=0), int8 coefficients only (DEJAVU_FP4=0);
#   mfma+fold : the matrix-core kernel as it ships: the fp4 form where the patches sit on the library's levels, the
#             finishing epilogue inside the kernel, the step ends in k_fold;
#   default : nothing set -- what ships (the engine times the kernel forms and picks the step ending by library size).
ENGINE_MODES = ["k_finish", "k_combine+k_tail", "mfma", "mfma+fold", "default"]
_MODE_ENV = {
    "k_finish": {"DEJAVU_FINISH": "2"},
    "k_combine+k_tail": {"DEJAVU_FINISH": "0"},
    "mfma": {"DEJAVU_FINISH": "2", "DEJAVU_SHAPE": "6", "DEJAVU_BITS": "2", "DEJAVU_FUSE": "0", "DEJAVU_FP4": "0"},
    "mfma+fold": {"DEJAVU_SHAPE": "6", "DEJAVU_BITS": "2"},
    "default": {},
}
_MODE_KEYS = ("DEJAVU_FINISH", "DEJAVU_SHAPE", "DEJAVU_BITS", "DEJAVU_FENCED", "DEJAVU_FUSE", "DEJAVU_FP4")


class engine_mode(object):
    """Context manager: the environment of one engine mode while an engine is created."""

    def __init__(self, mode):
        self.env = _MODE_ENV[mode]

    def __enter__(self):
        import os
        self.before = {k: os.environ.get(k) for k in _MODE_KEYS}
        for k in _MODE_KEYS:
            os.environ.pop(k, None)
        os.environ.update(self.env)
        return self

    def __exit__(self, *exc):
        import os
        for k, v in self.before.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
        return False
