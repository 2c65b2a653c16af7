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
