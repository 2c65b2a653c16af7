"""The oracle (oracle/sads_oracle.c) against the reference's own outputs: bit-for-bit.

These fixtures were produced by tests/golden/make_golden.py running the reference's Cython
kernel (navsim/util.pyx:31-73) and agent loop (navsim/NavBySceneFamiliarity.py:279-329).
"""
import numpy as np
import pytest

from oracle import oracle
from tests.helpers import kernel_case_inputs, step_case_inputs


def test_kernel_vectors_bit_exact(manifest, golden):
    z = golden("t1_kernel.npz")
    assert len(manifest["t1_kernel"]) == 24
    for case in manifest["t1_kernel"]:
        lib, scene = kernel_case_inputs(case)
        fam = oracle.sads_hsv(lib, scene, case["chem_weight"])
        ref = z[case["key"]]
        assert fam.tobytes() == ref.tobytes(), case["key"]
        func = oracle.sads_familiarity(case["chem_weight"])(lib)
        assert func.max_familiarity == case["max_familiarity"]


def test_step_vectors_bit_exact(manifest, golden):
    z = golden("t2_step.npz")
    for case in manifest["t2_step"]:
        lib, patches = step_case_inputs(case)
        r = oracle.step(lib, patches, case["chem_weight"])
        assert r["angle_familiarity"].tobytes() == z[case["name"] + "_angle"].tobytes(), case["name"]
        assert r["scene_familiarity"].tobytes() == z[case["name"] + "_scene"].tobytes(), case["name"]
        assert r["best_idex"] == case["best_idex"], case["name"]
        assert r["best_view"] == case["best_view"], case["name"]
        assert r["step_familiarity"] == case["step_familiarity"], case["name"]


def test_ssds_bit_exact(golden):
    z = golden("t6_ssds.npz")
    assert oracle.ssds(z["a"], z["b"]) == float(z["ssd"])


def test_known_answers():
    """Hand-derivable values (tier T0)."""
    h, w = 6, 10
    base = np.zeros((3, h, w, 3), dtype=np.uint8)
    base[0, ..., 2] = 255                      # V all 255
    base[1, ..., 0] = 9; base[1, ..., 1] = 255  # hue 9, S 255, V 0
    base[2, ..., 0] = 7; base[2, ..., 1] = 0    # hue 7, S 0
    scene0 = np.zeros((h, w, 3), dtype=np.uint8)
    # identical scene -> h*w exactly; V 0 vs 255 with cw=0 -> sum of exact 1.0 -> 0.0
    fam = oracle.sads_hsv(base, scene0, 0.0)
    assert fam[0] == 0.0 and fam[1] == h * w and fam[2] == h * w
    # different hue, S=255 both, cw=1: (255+255)*0.5/255 = 1.0 per pixel -> 0.0
    scene = np.zeros((h, w, 3), dtype=np.uint8)
    scene[..., 0] = 3; scene[..., 1] = 255
    fam = oracle.sads_hsv(base, scene, 1.0)
    assert fam[1] == 0.0
    # same hue, S 255 vs 0, cw=1: 0.5 per pixel
    scene[..., 0] = 7
    fam = oracle.sads_hsv(base, scene, 1.0)
    assert fam[2] == h * w - 0.5 * h * w
    # dtype errors surface as ValueError like the reference's buffer mismatch
    with pytest.raises(ValueError):
        oracle.sads_hsv(base.astype(np.float32), scene0, 0.0)


def test_dense_synthetic_oracle_is_pinned_to_the_integer_sums():
    """oracle_synth_int_sums (the full-size dense check of tests/test_gpu_parity.py regenerates every view inside its C loop)
    gives exactly oracle_int_sums on the views navsim_amd.synth.synth_views makes -- both saturation flavours, a first_view
    that is not zero, a sensor whose pixel count is not a multiple of anything."""
    if not oracle.have_omp():
        pytest.skip("oracle/liboracle_omp.so is not built")
    from navsim_amd import synth
    for full in (False, True):
        for seed, first, F, h, w in ((5, 1000, 300, 9, 13), (777, 499990, 10, 128, 128)):
            lib = synth.synth_views(seed, F, h, w, first_view=first, full_range_s=full)
            scene = synth.synth_patches(seed, 1, h, w, full_range_s=full)[0]
            want = oracle.int_sums(lib, scene)
            got = oracle.synth_int_sums(seed, first, F, h, w, scene, full_range_s=full, threads=3)
            assert np.array_equal(want[0], got[0]) and np.array_equal(want[1], got[1])
            fam = oracle.fam_from_sums(got[0], got[1], h * w, 0.25)
            np.testing.assert_allclose(fam, oracle.sads_hsv(lib, scene, 0.25), rtol=1e-12)
