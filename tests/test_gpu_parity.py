"""GPU parity tests: the HIP path (through the C ABI) against the oracle and the golden fixtures.

Bar: heading / view indices bit-exact; integer-sum scores within 1e-9 relative of the reference's
double (observed ~1e-13); exact mode bit-identical.  Everything here calls libdejavu_hip.so via
navsim_amd (ctypes); nothing reads /root/reference.
"""
import numpy as np
import pytest

import navsim_amd
from navsim_amd import synth
from oracle import oracle
from tests.helpers import ENGINE_MODES, engine_mode, kernel_case_inputs, step_case_inputs
from tests.test_host_logic import check_trajectory

pytestmark = pytest.mark.gpu

RTOL = 1e-9


@pytest.fixture(scope="module", params=ENGINE_MODES)
def eng(request):
    """Every test runs under each form of the scoring path (tests/helpers.py:ENGINE_MODES): both step endings, the
    bit-plane matrix-core kernel, and the product default."""
    with engine_mode(request.param):
        e = navsim_amd.FamiliarityEngine(device=0)
    e.mode = request.param
    yield e
    e.close()


def expected_planes(lib, info):
    """NumPy model of the device layout's stored bytes: uint8[F, n_planes, P]."""
    F = lib.shape[0]
    flat = lib.reshape(F, -1, 3)
    planes = []
    if info["generic_hue"]:
        planes += [flat[..., 0], flat[..., 1]]
    elif info["signed_saturation"]:
        h0, h1 = info["hues"]
        x = np.where(flat[..., 0] == h0, flat[..., 1].astype(int), 0) - np.where(flat[..., 0] == h1, flat[..., 1].astype(int), 0)
        planes.append((128 + x).astype(np.uint8))
    elif info["chem_weight"] > 0:
        for hue in info["hues"]:
            planes.append(np.where(flat[..., 0] == hue, flat[..., 1], 0).astype(np.uint8))
    if info["has_value_plane"]:
        planes.append(flat[..., 2])
    return np.stack(planes, axis=1)


@pytest.mark.parametrize("cw", [0.0, 0.4, 1.0])
def test_layout_round_trip(eng, cw):
    lib = synth.synth_views(5, 130, 5, 7)
    eng.set_library(lib, cw)
    info = eng.library_info()
    assert info["n_views"] == 130 and (info["h"], info["w"]) == (5, 7)
    assert info["n_planes"] == (0 if cw == 0 else 1) + (0 if cw == 1 else 1)       # S <= 127, two hues: one signed plane
    if cw > 0:
        assert info["hues"] == [0, 127] and not info["generic_hue"] and info["signed_saturation"]
    assert np.array_equal(eng.read_planes(0, 130), expected_planes(lib, info))
    assert np.array_equal(eng.read_planes(64, 3), expected_planes(lib, info)[64:67])


def test_generated_library_equals_uploaded(eng):
    F, h, w = 300, 9, 11
    for cw in (0.0, 0.3):
        eng.generate_library(77, F, h, w, cw, first_view=1000)
        gen = eng.read_planes(0, F)
        lib = synth.synth_views(77, F, h, w, first_view=1000)
        eng.set_library(lib, cw, first_view=1000)
        assert np.array_equal(gen, eng.read_planes(0, F))


def test_kernel_golden_vectors(eng, manifest, golden):
    z = golden("t1_kernel.npz")
    for case in manifest["t1_kernel"]:
        lib, scene = kernel_case_inputs(case)
        ref = z[case["key"]]
        eng.set_exact(False)
        eng.set_library(lib, case["chem_weight"])
        fam = np.full(case["F"], np.nan)
        eng.score(scene, fam)
        np.testing.assert_allclose(fam, ref, rtol=RTOL, atol=0, err_msg=case["key"])
        eng.set_exact(True)
        eng.score(scene, fam)
        assert fam.tobytes() == ref.tobytes(), case["key"]          # exact mode: bit for bit
    eng.set_exact(False)


def test_plugin_factory_contract():
    lib = synth.synth_views(9, 70, 6, 6)
    with pytest.raises(AssertionError):
        navsim_amd.sads_familiarity(1.5)(lib)                       # util.pyx:12
    func = navsim_amd.sads_familiarity(0.25)(lib)
    assert func.max_familiarity == 36
    fam = np.empty(70)
    func(lib[3], fam)
    assert fam[3] == 36.0 and np.argmax(fam) == 3
    with pytest.raises(ValueError):
        func(lib[3].astype(np.float32), fam)                        # "Buffer dtype mismatch"
    with pytest.raises(ValueError):
        func(lib[3], np.empty(70, dtype=np.float32))
    with pytest.raises(ValueError):
        navsim_amd.sads_familiarity(0.0)(lib.astype(np.float32))
    func.engine.close()


def test_step_golden_vectors(eng, manifest, golden):
    z = golden("t2_step.npz")
    for case in manifest["t2_step"]:
        lib, patches = step_case_inputs(case)
        eng.set_library(lib, case["chem_weight"])
        for exact in (False, True):
            eng.set_exact(exact)
            r = eng.step(patches, want_scene=True)
            name = case["name"]
            assert r["best_idex"] == case["best_idex"], (name, exact, r["n_candidates"], r["flags"])
            assert r["best_view"] == case["best_view"], (name, exact)
            np.testing.assert_allclose(r["angle_familiarity"], z[name + "_angle"], rtol=RTOL, err_msg=name)
            np.testing.assert_allclose(r["scene_familiarity"], z[name + "_scene"], rtol=RTOL, err_msg=name)
            np.testing.assert_allclose(r["step_familiarity"], case["step_familiarity"], rtol=RTOL)
            if exact or (r["flags"] & 1):
                assert r["step_familiarity"] == case["step_familiarity"], name   # resolved values are exact
            if exact:
                assert r["angle_familiarity"].tobytes() == z[name + "_angle"].tobytes()
                assert r["scene_familiarity"].tobytes() == z[name + "_scene"].tobytes()
    eng.set_exact(False)


def test_tie_cases_take_the_resolver(eng, manifest):
    """The tie-stress fixtures really exercise the exact resolver / overflow path."""
    seen = {}
    for case in manifest["t2_step"]:
        if not case["name"].startswith(("s_ties", "s_dup")):
            continue
        lib, patches = step_case_inputs(case)
        eng.set_library(lib, case["chem_weight"])
        r = eng.step(patches, want_scene=False)
        seen[case["name"]] = (r["n_candidates"], r["flags"])
        assert r["best_idex"] == case["best_idex"]
    assert any(f & 1 for _, f in seen.values()), seen        # resolver ran somewhere
    assert seen["s_dup"][0] > 1000 and seen["s_dup"][1] & 1, seen


def test_candidate_overflow_falls_back_to_exact(eng):
    """More near-ties than the candidate list holds: the step is redone with exact scores."""
    base = synth.synth_views(3, 1, 8, 8)[0]
    lib = np.repeat(base[None], 1000, axis=0)
    lib[500, 2, 2, 2] ^= 0x40
    patches = np.repeat(base[None], 8, axis=0)
    patches[:, 0, 0, 2] ^= 0x80
    patches[3, 7, 7, 2] ^= 0x01
    for cw in (0.0, 0.5):
        eng.set_library(lib, cw)
        r = eng.step(patches, want_scene=True)
        want = oracle.step(lib, patches, cw)
        assert r["flags"] & 4 and r["flags"] & 2, r["flags"]
        assert r["n_candidates"] > 4096
        assert r["best_idex"] == want["best_idex"] and r["best_view"] == want["best_view"]
        assert r["angle_familiarity"].tobytes() == want["angle_familiarity"].tobytes()
        assert r["scene_familiarity"].tobytes() == want["scene_familiarity"].tobytes()


SHAPES = [
    # F, h, w, A, cw, data
    (1, 1, 1, 1, 0.0, "levels"),
    (63, 3, 5, 3, 0.5, "levels"),
    (65, 4, 4, 16, 1.0, "levels"),
    (200, 7, 9, 17, 0.25, "levels"),
    (129, 16, 16, 33, 0.7, "levels"),
    (100, 10, 6, 60, 0.0, "levels"),
    (90, 8, 8, 64, 0.3, "levels"),
    (150, 6, 11, 5, 0.5, "manyhues"),
    (80, 8, 8, 20, 1.0, "manyhues"),
    (70, 5, 5, 4, 0.0, "manyhues"),
    (120, 9, 9, 8, 0.6, "threehues"),
    (64, 12, 12, 8, 0.5, "foreignhue"),
    (64, 12, 12, 8, 1.0, "zerosat"),
    (150, 9, 13, 12, 0.35, "signed"),
    (90, 16, 16, 16, 1.0, "signed"),
    (70, 8, 8, 5, 0.5, "twohues_big_s"),
    # many saturation values, few value levels: the mixed layout (value bit planes on the matrix cores + saturation bytes)
    (900, 16, 16, 12, 0.35, "mixed_signed"),
    (5000, 16, 16, 33, 0.3, "mixed_signed"),
    (700, 12, 20, 32, 0.5, "mixed_threehues"),
    (600, 16, 16, 9, 0.25, "mixed_offv"),
    (2100, 9, 13, 5, 1.0, "mixed_signed"),
]


def make_inputs(F, h, w, A, kind, seed):
    if kind == "levels":
        return synth.synth_views(seed, F, h, w), synth.synth_patches(seed, A, h, w)
    lib = synth.random_hsv(seed, (F, h, w, 3))
    pat = synth.random_hsv(seed + 1, (A, h, w, 3))
    if kind.startswith("mixed"):     # value in five levels, saturation in many
        lib[..., 2] = synth.V_LEVELS[lib[..., 2] % 5]
        pat[..., 2] = synth.V_LEVELS[pat[..., 2] % 5] if kind != "mixed_offv" else pat[..., 2]
        if kind == "mixed_threehues":
            lib[..., 0] = (lib[..., 0] % 3) * 40
            pat[..., 0] = (pat[..., 0] % 4) * 40
        else:                        # two hues, S <= 127: one signed saturation plane with ~255 values
            lib[..., 0] = np.where(lib[..., 0] & 1, 200, 10)
            lib[..., 1] >>= 1
            pat[..., 0] = np.where(pat[..., 0] & 1, 200, 10)
        return lib, pat
    if kind == "manyhues":           # > 4 hues with S > 0: generic-hue layout
        lib[..., 0] &= 0x0F
        pat[..., 0] &= 0x0F
    elif kind == "threehues":        # one-hot layout with 3 planes
        lib[..., 0] = (lib[..., 0] % 3) * 40
        pat[..., 0] = (pat[..., 0] % 3) * 40
    elif kind == "foreignhue":       # patch hues the library never uses -> per-heading constant
        lib[..., 0] = (lib[..., 0] % 2) * 9
        pat[..., 0] = (pat[..., 0] % 4) * 9
    elif kind == "zerosat":          # library without any saturation
        lib[..., 1] = 0
    elif kind == "signed":           # two hues, library S <= 127 -> one signed plane; patches exceed it and bring a third hue
        lib[..., 0] = np.where(lib[..., 0] & 1, 200, 10)
        lib[..., 1] >>= 1
        pat[..., 0] = np.choose(pat[..., 0] % 3, [10, 200, 77])
    elif kind == "twohues_big_s":    # two hues but S up to 255 -> stays two one-hot planes
        lib[..., 0] = np.where(lib[..., 0] & 1, 200, 10)
        pat[..., 0] = np.where(pat[..., 0] & 1, 200, 10)
    return lib, pat


@pytest.mark.parametrize("F,h,w,A,cw,kind", SHAPES)
def test_ragged_shapes_against_oracle(eng, F, h, w, A, cw, kind):
    lib, pat = make_inputs(F, h, w, A, kind, seed=F * 131 + A)
    eng.set_library(lib, cw)
    info = eng.library_info()
    if kind == "manyhues" and cw > 0:
        assert info["generic_hue"]
    if kind == "threehues" and cw > 0:
        assert info["n_hue_planes"] == 3
    if kind == "signed":
        assert info["signed_saturation"] and info["n_hue_planes"] == 1
    if kind == "twohues_big_s":
        assert not info["signed_saturation"] and info["n_hue_planes"] == 2
    if kind.startswith("mixed") and eng.mode == "mfma+fold" and 0.0 < cw < 1.0:
        assert info["mixed_layout"] and info["bit_planes_hs"] == 0 and info["bit_planes_v"] == 4, info
    want = oracle.step(lib, pat, cw)
    for exact in (False, True):
        eng.set_exact(exact)
        got = eng.step(pat, want_scene=True)
        assert got["best_idex"] == want["best_idex"], (exact, got["n_candidates"], got["flags"])
        assert got["best_view"] == want["best_view"]
        if not exact:                                            # the step as the agent runs it (finished inside the scoring kernel where it can be)
            fused = eng.step(pat, want_scene=False)
            assert (fused["best_idex"], fused["best_view"]) == (want["best_idex"], want["best_view"])
            assert np.array_equal(fused["angle_familiarity"], got["angle_familiarity"]) and np.array_equal(fused["angle_view"], got["angle_view"])
            if kind.startswith("mixed") and eng.mode == "mfma+fold" and 0.0 < cw < 1.0:
                form = eng.scoring_form()
                assert form["matrix_cores"] and not form["fused_finish"], form      # two passes (value bits, saturation bytes) meet in k_finish
        np.testing.assert_allclose(got["angle_familiarity"], want["angle_familiarity"], rtol=RTOL, atol=1e-12)
        np.testing.assert_allclose(got["scene_familiarity"], want["scene_familiarity"], rtol=RTOL, atol=1e-12)
        if exact:
            assert got["angle_familiarity"].tobytes() == want["angle_familiarity"].tobytes()
            assert got["scene_familiarity"].tobytes() == want["scene_familiarity"].tobytes()
        fam = np.empty(F)
        eng.score(pat[A // 2], fam)
        ref = oracle.sads_hsv(lib, pat[A // 2], cw)
        if exact:
            assert fam.tobytes() == ref.tobytes()
        else:
            np.testing.assert_allclose(fam, ref, rtol=RTOL, atol=1e-12)
    eng.set_exact(False)


def test_integer_sums_are_exact(eng):
    """cw=0 and cw=1 scores are pure functions of the exact integer sums (util.pyx:48-56,69)."""
    lib, pat = make_inputs(257, 13, 9, 2, "threehues", 42)
    s_hs, s_v = oracle.int_sums(lib, pat[0])
    fam = np.empty(257)
    eng.set_library(lib, 0.0)
    eng.score(pat[0], fam)
    assert np.array_equal(fam, 13 * 9 - s_v / 255.0)
    eng.set_library(lib, 1.0)
    eng.score(pat[0], fam)
    assert np.array_equal(fam, 13 * 9 - (0.5 * s_hs) / 255.0)


def test_errors(eng):
    lib = synth.synth_views(1, 10, 4, 4)
    with pytest.raises(ValueError):
        eng.set_library(lib, 1.5)
    with pytest.raises(ValueError):
        eng.set_library(lib[..., :2], 0.0)
    eng.set_library(lib, 0.0)
    r = eng.step(np.zeros((65, 4, 4, 3), dtype=np.uint8))          # more than one library pass: the wide step (round 4)
    assert r["n_passes"] == 2 and r["best_idex"] == 0
    import ctypes
    from navsim_amd import _native as N
    big = np.zeros((65, 4, 4, 3), dtype=np.uint8)                     # dv_step itself still holds DV_MAX_HEADINGS at most
    assert eng._lib.dv_step(eng._ctx, N.u8ptr(big), 65, 0, ctypes.byref(N.StepResult()), None) == -1
    with pytest.raises(ValueError):
        eng.step(np.zeros((2, 5, 4, 3), dtype=np.uint8))
    eng.clear_library()
    with pytest.raises(navsim_amd.EngineError):
        eng.step(np.zeros((2, 4, 4, 3), dtype=np.uint8))


def test_trajectories_match_reference_on_gpu(manifest, golden):
    """1000-step trajectories through the product agent: sensor model, scoring and decision all on the GPU."""
    z = golden("t4_trajectory.npz")
    for case in manifest["t4_trajectory"]:
        check_trajectory(case, z, navsim_amd.sads_familiarity(case["chem_weight"]), fam_rtol=RTOL)


def test_trajectory_with_host_sensor_and_uploaded_patches(manifest, golden):
    """Same agent with the sensor model on the host (patches uploaded each step through dv_step)."""
    z = golden("t4_trajectory.npz")
    case = manifest["t4_trajectory"][2]
    check_trajectory(case, z, navsim_amd.sads_familiarity(case["chem_weight"]), fam_rtol=RTOL, use_gpu_sensor=False)


def test_gpu_sensor_model_matches_reference(manifest, golden):
    """k_sense against the reference's get_sensor_mat outputs (tests/golden/t5_sensor.npz)."""
    z = golden("t5_sensor.npz")
    meta = manifest["t5_sensor"]["meta"]
    land = synth.synth_landscape(meta["landscape"]["seed"], meta["landscape"]["size"], meta["landscape"]["grain"])
    lands = {"land": land, "land2": z["land2"]}
    for case in manifest["t5_sensor"]["cases"]:
        levels = case["n_sensor_levels"]
        levels = tuple(levels) if isinstance(levels, list) else levels
        nsf = navsim_amd.NavBySceneFamiliarity(
            lands[case["landscape"]], case["sensor_dimensions"], 1.0, n_test_angles=4,
            sensor_pixel_dimensions=case["sensor_pixel_dimensions"], n_sensor_levels=levels,
            mask_middle_n=case["mask_middle_n"], familiarity_model=navsim_amd.sads_familiarity())
        assert nsf._engine is not None
        for k, (x, y, a) in enumerate(case["poses"]):
            assert np.array_equal(nsf.get_sensor_mat((x, y), a), z[case["name"] + "_mats"][k]), (case["name"], k)
        nsf._engine.close()


@pytest.mark.parametrize("sdim,spd,levels,mask", [((32, 32), [1, 1], 5, 0), ((16, 8), [2, 4], 4, 1),
                                                  ((10, 6), [3, 5], (7, 256, 3), 2), ((64, 64), [1, 1], 5, 0)])
def test_gpu_sensor_equals_host_sensor_on_random_poses(golden, sdim, spd, levels, mask):
    """Many random poses: the GPU sensor model and the (fixture-pinned) host sensor model agree byte for byte."""
    land2 = golden("t5_sensor.npz")["land2"]
    land = np.tile(land2, (2, 2, 1))                       # 600 x 600
    gpu = navsim_amd.NavBySceneFamiliarity(land, sdim, 1.0, n_test_angles=4, sensor_pixel_dimensions=spd,
                                           n_sensor_levels=levels, mask_middle_n=mask,
                                           familiarity_model=navsim_amd.sads_familiarity())
    host = navsim_amd.NavBySceneFamiliarity(land, sdim, 1.0, n_test_angles=4, sensor_pixel_dimensions=spd,
                                            n_sensor_levels=levels, mask_middle_n=mask, use_gpu_sensor=False,
                                            familiarity_model=oracle.sads_familiarity())
    rng = np.random.default_rng(7)
    n = 60
    xs, ys = rng.uniform(120, 480, n), rng.uniform(120, 480, n)
    xs[:8] = np.round(xs[:8]) + 0.5                        # exact .5 coordinates: C round() half away from zero
    ys[:8] = np.round(ys[:8]) - 0.5
    angs = rng.uniform(0, 2 * np.pi, n)
    angs[:4] = [0.0, np.pi / 2, np.pi, 3 * np.pi / 2]
    got = gpu._engine.sense(xs, ys, angs)
    for i in range(n):
        assert np.array_equal(got[i], host.get_sensor_mat((xs[i], ys[i]), angs[i])), i
    gpu._engine.close()


def test_gpu_sensor_index_errors_like_the_reference():
    land = synth.synth_landscape(3, 120, 4)
    nsf = navsim_amd.NavBySceneFamiliarity(land, (40, 40), 1.0, n_test_angles=4,
                                           familiarity_model=navsim_amd.sads_familiarity())
    host = navsim_amd.NavBySceneFamiliarity(land, (40, 40), 1.0, n_test_angles=4, use_gpu_sensor=False,
                                            familiarity_model=oracle.sads_familiarity())
    # corners of a rotated sensor reach r*sqrt(2): negative indices wrap, indices past the end raise IndexError
    near_origin = (20.5, 20.5)
    assert np.array_equal(nsf.get_sensor_mat(near_origin, 0.8), host.get_sensor_mat(near_origin, 0.8))
    far_corner = (99.4, 99.4)
    with pytest.raises(IndexError):
        host.get_sensor_mat(far_corner, 0.8)
    with pytest.raises(IndexError):
        nsf.get_sensor_mat(far_corner, 0.8)
    with pytest.raises(navsim_amd.OutOfLandscapeBoundsException):
        nsf.get_sensor_mat((10.0, 60.0), 0.0)
    # the same condition inside a fused step: reported when the step is waited for
    path = np.stack([np.linspace(40, 80, 30), np.full(30, 60.0)], axis=1)
    nsf.train_from_path(path)
    nsf.position, nsf.angle = far_corner, 0.8 - nsf.angle_offsets[0]
    with pytest.raises(IndexError):
        nsf.step_forward(fake=True)
    nsf.position, nsf.angle = (60.0, 60.0), 0.0
    nsf.step_forward(fake=True)                      # and the engine is usable afterwards
    nsf._engine.close()


def _check_shipped_step_at_size(eng, seed, F, h, w, A, cw, patches, r_unfused, require_form=True, plain=True):
    """The step as it ships -- want_scene=False, no force_resolve: the matrix-core kernel in its fp4 form, scores finished
    in its epilogue, k_fold -- against (1) the unfused step (k_finish on the partial sums) bit for bit: every per-heading
    maximum, its view, the decision; (2) the oracle on the reported views plus a fixed spread of the library: the sample
    contains every heading's maximiser, so the oracle's per-heading maxima, views and decision on it must be the reported
    ones, and no sampled view may beat them."""
    r = eng.step(patches, want_scene=False)
    form = eng.scoring_form()
    if require_form:
        assert form["matrix_cores"] and form["fp4"] and form["fused_finish"], form
    if plain:
        assert not (r["flags"] & 1), "a planted unique best must not need the exact resolver"
    assert np.array_equal(r["angle_familiarity"], r_unfused["angle_familiarity"])
    assert np.array_equal(r["angle_view"], r_unfused["angle_view"])
    assert (r["best_idex"], r["best_view"]) == (r_unfused["best_idex"], r_unfused["best_view"])
    assert r["step_familiarity"] == r_unfused["step_familiarity"]
    views = np.unique(np.concatenate([np.arange(0, F, max(1, F // 97)), np.asarray(r["angle_view"], dtype=np.int64)]))
    lib = np.stack([synth.synth_views(seed, 1, h, w, first_view=int(f))[0] for f in views])
    want = oracle.step(lib, patches, cw)
    np.testing.assert_allclose(r["angle_familiarity"], want["angle_familiarity"], rtol=RTOL)
    for a in range(A):                       # each heading's first maximiser on the sample is the reported view
        fam = oracle.sads_hsv(lib, patches[a], cw)
        assert int(views[int(np.argmax(fam))]) == int(r["angle_view"][a]), a
    assert r["best_idex"] == want["best_idex"] and r["best_view"] == int(views[want["best_view"]])
    return r


def test_full_size_properties():
    """BASELINE config 1 (64x64, 50k views, 16 headings) through size-independent properties."""
    F, h, w, A, seed = 50000, 64, 64, 16, 20261004
    eng = navsim_amd.FamiliarityEngine(device=0)
    for cw in (0.0, 0.25):
        eng.generate_library(seed, F, h, w, cw)
        patches = synth.synth_patches(seed, A, h, w)
        # plant near-copies of two far-apart views: the closer one must win, at its heading
        v1 = synth.synth_views(seed, 1, h, w, first_view=41234)[0]
        v2 = synth.synth_views(seed, 1, h, w, first_view=77)[0]
        patches[11] = synth.near_match_patch(v1, 5, fraction=0.01)
        patches[2] = synth.near_match_patch(v2, 6, fraction=0.05)
        r = eng.step(patches, want_scene=True)
        assert r["best_idex"] == 11 and r["best_view"] == 41234
        assert r["angle_view"][2] == 77
        assert r["step_familiarity"] > 0.98 * h * w
        _check_shipped_step_at_size(eng, seed, F, h, w, A, cw, patches, r, require_form=cw > 0)
        # sampled views against the oracle (regenerated on the host from the same seed)
        for f0 in (0, 41230, F - 8):
            sub = synth.synth_views(seed, 8, h, w, first_view=f0)
            want = oracle.step(sub, patches, cw)
            np.testing.assert_allclose(r["scene_familiarity"][f0:f0 + 8], want["scene_familiarity"], rtol=RTOL)
        # scene_familiarity is a lower bound of every heading's score; maxima are attained
        assert np.all(r["scene_familiarity"] <= r["angle_familiarity"].max() + 1e-9)
        # sharded == unsharded: two half libraries reproduce the per-heading maxima and views
        full_angle, full_view = r["angle_familiarity"], r["angle_view"]
        halves = []
        for lo, hi in ((0, F // 2), (F // 2, F)):
            eng.generate_library(seed, hi - lo, h, w, cw, first_view=lo)
            halves.append(eng.step(patches, want_scene=False))
        merged = np.maximum(halves[0]["angle_familiarity"], halves[1]["angle_familiarity"])
        assert np.array_equal(merged, full_angle)
        pick = np.where(halves[0]["angle_familiarity"] >= halves[1]["angle_familiarity"],
                        halves[0]["angle_view"], halves[1]["angle_view"])
        assert np.array_equal(pick, full_view)
    eng.close()


def test_large_library_properties():
    """BASELINE config 2 (128x128 sensor, 500k views, 32 headings; 16.4 GB of tiles) through size-independent properties."""
    F, h, w, A, seed, cw = 500000, 128, 128, 32, 777, 0.25
    eng = navsim_amd.FamiliarityEngine(device=0)
    eng.generate_library(seed, F, h, w, cw)
    info = eng.library_info()
    assert info["n_planes"] == 2 and info["signed_saturation"]          # synth: two hues, S in {0,127}
    assert info["tile_bytes"] >= F * h * w * info["n_planes"]
    patches = synth.synth_patches(seed, A, h, w)
    targets = {5: 499999, 17: 250001, 30: 3}                  # heading -> planted view (first, middle, last groups)
    for a, f in targets.items():
        patches[a] = synth.near_match_patch(synth.synth_views(seed, 1, h, w, first_view=f)[0], 100 + a,
                                            fraction=0.01 * (1 + a % 3))
    r = eng.step(patches, want_scene=True)
    for a, f in targets.items():
        assert r["angle_view"][a] == f, (a, r["angle_view"][a])
    assert r["best_idex"] == 30 and r["best_view"] == 3       # 1 % perturbed beats 2 % and 3 %
    # sampled views against the oracle
    for f0 in (0, 250000, F - 4):
        sub = synth.synth_views(seed, 4, h, w, first_view=f0)
        want = oracle.step(sub, patches, cw)
        np.testing.assert_allclose(r["scene_familiarity"][f0:f0 + 4], want["scene_familiarity"], rtol=RTOL)
    _check_shipped_step_at_size(eng, seed, F, h, w, A, cw, patches, r)
    # fresh patches through the on-device generator (what bench.py times), same check
    eng.generate_patches(seed + 1, A)
    fresh = synth.synth_patches(seed + 1, A, h, w)
    eng.step_enqueue(want_scene=False)
    r_gen = eng.step_wait(want_scene=False)
    r_up = eng.step(fresh, want_scene=True)
    assert np.array_equal(r_gen["angle_familiarity"], r_up["angle_familiarity"]) and np.array_equal(r_gen["angle_view"], r_up["angle_view"])
    _check_shipped_step_at_size(eng, seed, F, h, w, A, cw, fresh, r_up, plain=False)
    # exact value of the winner is reproduced by the resolver when forced
    r2 = eng.step(patches, want_scene=False, force_resolve=True)
    win = synth.synth_views(seed, 1, h, w, first_view=3)
    assert r2["flags"] & 1 and r2["best_idex"] == 30
    assert r2["step_familiarity"] == oracle.sads_hsv(win, patches[30], cw)[0]
    eng.close()


@pytest.mark.parametrize("cw", [0.0, 0.4])
def test_batched_agents_match_per_agent_reference(eng, cw):
    """dv_step_batch: every agent of an ensemble gets the decision the reference would take for it alone."""
    F, h, w, A, n_agents = 700, 8, 8, 6, 27          # 27 agents x 6 headings -> 3 passes (10 + 10 + 7 agents)
    lib = synth.synth_views(31, F, h, w)
    patches = synth.synth_views(32, n_agents * A, h, w).reshape(n_agents, A, h, w, 3)
    patches[3, 2] = lib[600]                          # an exact match for one agent
    patches[5, :] = patches[5, 0]                     # one agent whose headings all tie
    patches[11, 4] = synth.near_match_patch(lib[17], 1, fraction=0.05)
    eng.set_library(lib, cw)
    for exact in (False, True):
        eng.set_exact(exact)
        got = eng.step_batch(patches)
        assert len(got) == n_agents
        for i in range(n_agents):
            want = oracle.step(lib, patches[i], cw)
            assert got[i]["best_idex"] == want["best_idex"], (i, exact, got[i]["n_candidates"], got[i]["flags"])
            assert got[i]["best_view"] == want["best_view"], (i, exact)
            np.testing.assert_allclose(got[i]["angle_familiarity"], want["angle_familiarity"], rtol=RTOL)
        assert got[3]["best_idex"] == 2 and got[3]["best_view"] == 600 and got[3]["step_familiarity"] == h * w
    eng.set_exact(False)
    # a batch of one agent is the single-agent step
    one = eng.step_batch(patches[:1])[0]
    ref = eng.step(patches[0], want_scene=False)
    assert one["best_idex"] == ref["best_idex"] and np.array_equal(one["angle_familiarity"], ref["angle_familiarity"])


def _ssd_reference(lib, patches):
    """ssds() of the reference (navsim/util.pyx:171-184, via the pinned oracle) on the upcast data."""
    F, A = lib.shape[0], patches.shape[0]
    out = np.empty((A, F))
    lib64 = lib.astype(np.float64)
    for a in range(A):
        p64 = patches[a].astype(np.float64)
        for f in range(F):
            out[a, f] = oracle.ssds(p64, lib64[f])
    return out


@pytest.mark.parametrize("F,h,w,A", [(1, 1, 1, 1), (130, 5, 7, 3), (300, 16, 16, 16), (64, 9, 31, 8), (257, 32, 32, 10),
                                     (200, 12, 12, 32), (150, 10, 14, 64), (90, 7, 9, 37)])
def test_ssd_f32_metric_against_reference_ssds(eng, F, h, w, A):
    rng = np.random.default_rng(F * 7 + A)
    lib = rng.uniform(-3, 3, (F, h, w)).astype(np.float32)
    patches = rng.uniform(-3, 3, (A, h, w)).astype(np.float32)
    if F > 100:
        patches[A // 2] = lib[F // 3] + rng.normal(0, 0.01, (h, w)).astype(np.float32)    # a near match
    want = _ssd_reference(lib, patches)
    eng.set_library_f32(lib)
    for exact in (False, True):
        eng.set_exact(exact)
        r = eng.step_f32(patches, want_scene=True)
        best_a = int(np.argmin(want.min(axis=1)))
        assert r["best_idex"] == best_a and r["best_view"] == int(np.argmin(want[best_a])), (exact, r["flags"])
        np.testing.assert_allclose(r["angle_ssd"], want.min(axis=1), rtol=1e-6, atol=1e-12)      # north star: 1e-6 relative
        np.testing.assert_allclose(r["scene_ssd"], want.max(axis=0), rtol=1e-6, atol=1e-12)
        if exact:
            assert np.array_equal(r["angle_ssd"], want.min(axis=1))
        buf = eng.score_f32(patches[0])
        np.testing.assert_allclose(buf, want[0], rtol=1e-6 if not exact else 0, atol=1e-12 if not exact else 0)
    eng.set_exact(False)
    with pytest.raises(ValueError):
        eng.set_library_f32(lib.astype(np.float64))
    with pytest.raises((ValueError, navsim_amd.EngineError)):
        eng.step(np.zeros((2, h, w, 3), dtype=np.uint8))            # uint8 entry point on an f32 library


def test_ssd_f32_ties_and_duplicates(eng):
    rng = np.random.default_rng(5)
    base = rng.uniform(0, 1, (8, 8)).astype(np.float32)
    lib = np.repeat(base[None], 600, axis=0)
    lib[100] += np.float32(0.5)
    patches = np.repeat(base[None], 9, axis=0)
    patches[:, 0, 0] += np.float32(0.25)
    patches[4, 3, 3] -= np.float32(0.125)
    want = _ssd_reference(lib, patches)
    eng.set_library_f32(lib)
    r = eng.step_f32(patches)                       # 9 x 599 exact ties > candidate cap -> exact fallback
    best_a = int(np.argmin(want.min(axis=1)))
    assert r["best_idex"] == best_a and r["best_view"] == int(np.argmin(want[best_a]))
    assert r["flags"] & 4
    lib2 = rng.uniform(0, 1, (200, 8, 8)).astype(np.float32)
    lib2[150] = lib2[20]                            # two identical views: the first one wins
    eng.set_library_f32(lib2)
    p = np.stack([lib2[20], lib2[150] + np.float32(1e-3)])
    r = eng.step_f32(p)
    assert r["best_idex"] == 0 and r["best_view"] == 20 and r["step_ssd"] == 0.0


def _ssd_u8_reference(lib, patches):
    """[A, F] exact integer SSDs: the oracle's ssds (navsim/util.pyx:171-184) on the float64 upcast where the case is small enough,
    int64 arithmetic (equal to it: every partial sum of ssds is an integer below 2^53) everywhere."""
    l = lib.reshape(len(lib), -1).astype(np.int64)
    p = patches.reshape(len(patches), -1).astype(np.int64)
    want = np.stack([((l - p[a]) ** 2).sum(axis=1) for a in range(len(p))]).astype(np.float64)
    for a, f in ((0, 0), (len(p) - 1, len(l) - 1), (len(p) // 2, len(l) // 3)):
        assert oracle.ssds(patches[a].astype(np.float64), lib[f].astype(np.float64)) == want[a, f]
    return want


@pytest.mark.parametrize("F,h,w,A", [(1, 1, 1, 1), (31, 5, 7, 3), (33, 8, 4, 32), (700, 12, 20, 33), (2049, 32, 32, 16), (5000, 64, 64, 64),
                                      (300, 70, 67, 9), (1000, 100, 50, 40)])
def test_ssd_u8_metric_is_exact(eng, F, h, w, A):
    """ssd_u8 (int8 matrix cores, k_ssd_u8_mfma): every score equals the reference's ssds on the same uint8 data bit for bit,
    ragged sizes (pixels not a multiple of 32, views not of 32, 1..64 headings = one and two passes, more than one LDS chunk of
    K-steps at 70x67 and 100x50), extreme bytes included; the decision is the first heading, then the first view, of the minimum."""
    rng = np.random.default_rng(F * 11 + A)
    lib = rng.integers(0, 256, (F, h, w), dtype=np.uint8)
    patches = rng.integers(0, 256, (A, h, w), dtype=np.uint8)
    lib[F // 2] = 255
    patches[0] = 0                                            # the largest differences there are
    if F > 100:
        patches[A // 2] = lib[F // 3]
        patches[A // 2, 0, 0] ^= 1                           # a near match: SSD 1
    want = _ssd_u8_reference(lib, patches)
    eng.set_library_u8(lib)
    r = eng.step_u8(patches, want_scene=True)
    best_a = int(np.argmin(want.min(axis=1)))
    assert r["best_idex"] == best_a and r["best_view"] == int(np.argmin(want[best_a]))
    assert np.array_equal(np.array(r["angle_ssd"]), want.min(axis=1))
    assert [int(v) for v in r["angle_view"]] == [int(np.argmin(want[a])) for a in range(A)]
    assert np.array_equal(r["scene_ssd"], want.max(axis=0))
    assert r["step_ssd"] == want.min()
    for a in (0, A - 1):
        assert np.array_equal(eng.score_u8(patches[a]), want[a])
    with pytest.raises(ValueError):
        eng.set_library_u8(lib.astype(np.float32))
    with pytest.raises((ValueError, navsim_amd.EngineError)):
        eng.step(np.zeros((2, h, w, 3), dtype=np.uint8))            # sads_hsv entry point on an ssd_u8 library
    with pytest.raises((ValueError, navsim_amd.EngineError)):
        eng.step_f32(np.zeros((2, h, w), dtype=np.float32))


def test_ssd_u8_ties_go_to_the_first_heading_and_view(eng):
    rng = np.random.default_rng(8)
    lib = rng.integers(0, 256, (900, 9, 11), dtype=np.uint8)
    lib[700] = lib[40]
    lib[41] = lib[40]                                        # three identical views
    patches = rng.integers(0, 256, (12, 9, 11), dtype=np.uint8)
    patches[7] = lib[700]
    patches[3] = lib[41]                                     # two headings at SSD 0: heading 3 and view 40 win
    eng.set_library_u8(lib)
    r = eng.step_u8(patches)
    assert (r["best_idex"], r["best_view"], r["step_ssd"]) == (3, 40, 0.0)
    assert int(r["angle_view"][7]) == 40
    lib[:] = lib[0]                                          # a library of duplicates: every heading ties on every view
    eng.set_library_u8(lib)
    r = eng.step_u8(patches)
    want = _ssd_u8_reference(lib, patches)
    assert r["best_idex"] == int(np.argmin(want[:, 0])) and r["best_view"] == 0


def test_ssd_u8_largest_patch_and_extreme_bytes(eng):
    """The int32 cross terms at their limit: 131 071 pixels of bytes 0 (a' = -128 everywhere: the largest products) and 255, thirty
    LDS chunks of K-steps; one pixel more is refused."""
    h, w = 131071, 1
    lib = np.zeros((40, h, w), dtype=np.uint8)
    lib[7] = 255
    lib[9, ::2] = 255
    patches = np.zeros((3, h, w), dtype=np.uint8)
    patches[1] = 255
    patches[2, 1::2] = 255
    eng.set_library_u8(lib)
    r = eng.step_u8(patches, want_scene=True)
    want = _ssd_u8_reference(lib, patches)
    assert np.array_equal(np.array(r["angle_ssd"]), want.min(axis=1)) and np.array_equal(r["scene_ssd"], want.max(axis=0))
    assert want.max() == 131071 * 255.0 ** 2 and (r["best_idex"], r["best_view"]) == (0, 0)
    for a in range(3):
        assert np.array_equal(eng.score_u8(patches[a]), want[a])
    with pytest.raises((ValueError, navsim_amd.EngineError)):
        eng.set_library_u8(np.zeros((2, 131072, 1), dtype=np.uint8))


def test_ssd_u8_full_size_properties():
    """ssd_u8 on BASELINE configs[1]'s shape (64x64, 50 000 views; 16 and 64 headings): planted copies win at their headings with
    the exact SSDs of the reference on the planted pairs and on sampled views, and complementing every byte (x -> 255 - x) of both
    operands leaves every SSD unchanged."""
    F, h, w = 50000, 64, 64
    rng = np.random.default_rng(3)
    lib = rng.integers(0, 256, (F, h, w), dtype=np.uint8)
    eng = navsim_amd.FamiliarityEngine(0)
    try:
        eng.set_library_u8(lib)
        for A in (16, 64):
            patches = rng.integers(0, 256, (A, h, w), dtype=np.uint8)
            patches[A - 3] = lib[31337]
            patches[A - 3, 5, 5] ^= 0x10
            patches[2] = lib[49999]
            patches[2, :2] = 255 - patches[2, :2]
            r = eng.step_u8(patches, want_scene=True)
            assert (r["best_idex"], r["best_view"]) == (A - 3, 31337) and r["step_ssd"] == 256.0
            assert int(r["angle_view"][2]) == 49999
            for a, f in ((A - 3, 31337), (2, 49999), (0, 12345), (A - 1, 0)):
                want = oracle.ssds(patches[a].astype(np.float64), lib[f].astype(np.float64))
                if (a, f) in ((A - 3, 31337), (2, 49999)):
                    assert r["angle_ssd"][a] == want
                assert eng.score_u8(patches[a])[f] == want
            sample = rng.integers(0, F, 64)
            l = lib[sample].reshape(64, -1).astype(np.int64)
            p = patches.reshape(A, -1).astype(np.int64)
            want_max = np.stack([((l - p[a]) ** 2).sum(axis=1) for a in range(A)]).max(axis=0)
            assert np.array_equal(r["scene_ssd"][sample], want_max.astype(np.float64))
            if A == 16:
                first, p16 = np.array(r["angle_ssd"]), patches
        eng.set_library_u8(255 - lib)
        r = eng.step_u8(255 - p16)
        assert np.array_equal(np.array(r["angle_ssd"]), first) and (r["best_idex"], r["best_view"]) == (13, 31337)
    finally:
        eng.close()


@pytest.mark.parametrize("A", [5, 16, 30, 64])
def test_every_workgroup_shape_gives_identical_results(A):
    """The scoring kernel's forms (1 single-wave, 2 four waves + LDS sums, 3/4 heading ways, 5 packed accumulators;
    csrc/dejavu_hip.hip:launch_tiles_apad; 6 the bit-plane matrix-core kernel k_sad_mfma_dual) are normally chosen by timing; forced one by one they must produce the same
    integer sums, hence bit-identical scores and the reference's decision."""
    import os
    F, h, w, cw = 1500, 20, 24, 0.25
    lib = synth.synth_views(91, F, h, w)
    pats = synth.synth_patches(91, A, h, w)
    pats[A // 2] = synth.near_match_patch(lib[777], 5)
    want = oracle.step(lib, pats, cw)
    fams = {}
    try:
        for shape in (1, 2, 3, 4, 5, 6, 0):
            os.environ["DEJAVU_SHAPE"] = str(shape)
            e = navsim_amd.FamiliarityEngine(0)
            try:
                e.set_library(lib, cw)
                info = e.library_info()
                assert info["n_planes"] == 2 and info["signed_saturation"] == 1      # both sums live: shape 5 is valid
                assert info["has_bit_planes"] and (info["bit_planes_hs"], info["bit_planes_v"]) == (2, 4)   # shape 6 is
                r = e.step(pats, want_scene=False)
                assert (r["best_idex"], r["best_view"]) == (want["best_idex"], want["best_view"]), shape
                fams[shape] = np.array(r["angle_familiarity"])
                buf = np.empty(F)
                e.score(pats[0], buf)
                fams[(shape, "score")] = buf
            finally:
                e.close()
    finally:
        os.environ.pop("DEJAVU_SHAPE", None)
    for shape in (2, 3, 4, 5, 6, 0):
        np.testing.assert_array_equal(fams[shape], fams[1])
        np.testing.assert_array_equal(fams[(shape, "score")], fams[(1, "score")])
    np.testing.assert_allclose(fams[1], want["angle_familiarity"], rtol=1e-12)


def test_random_multi_block_ties(eng):
    """Duplicated views scattered over several k_finish blocks and repeated headings (tests/manual/stress_blocks.py's
    generator, 60 fixed cases): decisions, per-heading maxima and per-view minima against the oracle."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location(
        "stress_blocks", os.path.join(os.path.dirname(os.path.abspath(__file__)), "manual", "stress_blocks.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    rng = np.random.default_rng(2026)
    resolved = 0
    for it in range(60):
        lib, pat, cw = mod.make_case(rng)
        want = oracle.step(lib, pat, cw)
        eng.set_library(lib, cw)
        got = eng.step(pat, want_scene=True)
        assert (got["best_idex"], got["best_view"]) == (want["best_idex"], want["best_view"]), it
        np.testing.assert_allclose(got["angle_familiarity"], want["angle_familiarity"], rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(got["scene_familiarity"], want["scene_familiarity"], rtol=1e-9, atol=1e-12)
        resolved += bool(got["flags"] & 1)
    assert resolved >= 20


def test_workgroup_shape_is_reported_after_the_first_step(eng):
    lib = synth.synth_views(3, 700, 16, 16)
    pats = synth.synth_patches(3, 12, 16, 16)
    eng.set_library(lib, 0.25)
    if eng.mode in ("mfma", "mfma+fold"):                                   # DEJAVU_SHAPE=6: forced, nothing is timed
        assert eng.workgroup_shape(12) == 6 and eng.library_info()["has_bit_planes"]
        eng.step(pats, want_scene=False)
        assert eng.workgroup_shape(12) == 6
        return
    assert eng.workgroup_shape(12) == 0                      # nothing timed yet for this library
    eng.step(pats, want_scene=False)
    assert 1 <= eng.workgroup_shape(12) <= 6
    assert eng.workgroup_shape(40) == 0                      # another heading class: timed on its first use
    with pytest.raises(ValueError):                          # DV_ERR_INVALID maps to ValueError, as for the other calls
        eng.workgroup_shape(65)


@pytest.mark.parametrize("A", [64, 33, 16])
def test_ties_across_many_blocks(eng, A):
    """More k_finish blocks than its last block folds in one round (the two-pass walk over the block summaries), with
    the best view duplicated in far-apart blocks and seen under several headings."""
    F, h, w, cw = 70000, 4, 4, 0.25
    lib = synth.synth_views(17, F, h, w)
    pats = synth.synth_patches(17, A, h, w)
    star = lib[31000].copy()
    star[0, 0] = (77, 200, 13)                          # make it unlike the 2-hue, 5-level crowd: a unique best match
    for f in (69990, 31000, 45000, 300):
        lib[f] = star
    for a in (A - 1, A // 2, 3):
        pats[a] = star
    want = oracle.step(lib, pats, cw)
    assert (want["best_idex"], want["best_view"]) == (3, 300)
    eng.set_library(lib, cw)
    got = eng.step(pats, want_scene=True)
    assert (got["best_idex"], got["best_view"]) == (3, 300)
    assert got["n_candidates"] == 12 and got["flags"] & 1            # 4 views x 3 headings, settled by the exact resolver
    np.testing.assert_allclose(got["angle_familiarity"], want["angle_familiarity"], rtol=1e-12)
    np.testing.assert_allclose(got["scene_familiarity"], want["scene_familiarity"], rtol=1e-12)


def test_ensemble_matches_independent_agents():
    """navsim_amd.NavEnsemble (sensing + scoring of all agents batched per library pass, dv_sense_step_batch) against the
    same agents stepped one by one: identical headings, poses, familiarities, error metrics and stop codes."""
    land = synth.synth_landscape(11, 420, 4)
    path = synth.sin_training_path(0.5, 0.2 * 420, 0.6 * 420, arclen=1.0)[:260]
    kw = dict(n_test_angles=7, n_sensor_levels=5, saccade_degrees=120.0, max_distance_to_training_path=25.0,
              track_scene_familiarity=False)

    def trained():
        nsf = navsim_amd.NavBySceneFamiliarity(land, (16, 12), 1.0, familiarity_model=navsim_amd.sads_familiarity(0.25), **kw)
        nsf.train_from_path(path)
        return nsf

    d = path[2] - path[1]
    a0 = float(np.arctan2(d[1], d[0]) % (2 * np.pi))
    poses = [(path[1] + np.array([1.0, -0.5]), a0), (path[40] + np.array([-2.0, 1.0]), a0 + 0.2),
             (path[254] + np.array([0.0, 0.2]), float(np.arctan2(*(path[255] - path[254])[::-1]) % (2 * np.pi))),   # at the end of the path: reaches it
             (path[10] + np.array([0.0, 30.0]), a0),                  # far off the path: TooFar on its first step
             ((5.0, 5.0), 0.3),                                       # inside the bounds margin: OutOfLandscapeBounds
             (path[120] + np.array([3.0, -3.0]), a0 + 0.5), (path[60], a0), (path[61], a0), (path[62], a0), (path[63], a0)]
    # reference behaviour: each agent alone
    want = []
    for pos, ang in poses:
        nsf = trained()
        nsf.position, nsf.angle = (float(pos[0]), float(pos[1])), float(ang)
        row = dict(stop_status=0, completed_frames=0)
        try:
            for _ in range(70):
                nsf.step_forward()
                row["completed_frames"] += 1
        except navsim_amd.StopNavigationException as e:
            row["stop_status"] = e.get_code()
        if nsf._n_navigation_error:
            row["rmsd_error"], row["path_coverage"] = float(nsf.navigation_error), float(nsf.percent_recapitulated)
        want.append((row, nsf.position, nsf.angle, nsf.navigated_for_frames, nsf.angle_familiarity.copy()))
        nsf._engine.close()
    ens = navsim_amd.NavEnsemble.from_agent(trained(), poses)
    done = ens.run(70)
    codes = set()
    for i, (row, pos, ang, frames, afam) in enumerate(want):
        ag = ens.agents[i]
        assert ens.stop_status[i] == row["stop_status"], i
        assert done[i] == row["completed_frames"] and ag.navigated_for_frames == frames, i
        assert tuple(ag.position) == tuple(pos) and ag.angle == ang, i
        np.testing.assert_array_equal(ag.angle_familiarity, afam)
        if "rmsd_error" in row:
            assert float(ag.navigation_error) == row["rmsd_error"] and float(ag.percent_recapitulated) == row["path_coverage"]
        codes.add(row["stop_status"])
    assert codes == {0, 1, -1, -2}
    rows = navsim_amd.run_ensemble(ens, frames=0)                  # nothing left to run: just the result rows
    assert [r["stop_status"] for r in rows] == ens.stop_status and all("path_coverage" in r for r in rows)
    ens.engine.close()


def test_device_error_metrics_equal_the_host_arithmetic():
    """update_error on the device (dv_path_error_*, NavBySceneFamiliarity.py:252-276): nearest distances bit-equal to
    the reference's NumPy expression, the same coverage marks, answers in the order asked, and a finite
    max_distance_to_training_path stops the run inside the step as the reference does (:264)."""
    rng = np.random.default_rng(11)
    path = np.cumsum(rng.uniform(-1.5, 1.5, (70001, 2)), axis=0) + 500.0
    eng = navsim_amd.FamiliarityEngine(0)
    try:
        eng.set_training_path(path)
        reach = 2.4
        cov = np.zeros(len(path), dtype=bool)
        pos = path[rng.integers(0, len(path), 40)] + rng.uniform(-3, 3, (40, 2))
        want = []
        for k, (x, y) in enumerate(pos):
            eng.path_error_enqueue(x, y, reach)
            delta = path - (x, y)
            delta *= delta
            dist = np.sqrt(np.sum(delta, axis=1))
            want.append(np.min(dist))
            if want[-1] <= reach:
                cov |= dist <= reach
            if k % 5 == 4:                                   # collected late, several at a time, in order
                for j in range(k - 4, k + 1):
                    assert eng.path_error_wait() == want[j], j
        assert np.array_equal(eng.path_coverage(len(path)), cov) and cov.any()
        with pytest.raises(navsim_amd.EngineError):
            eng.path_error_wait()                            # nothing outstanding
        eng.path_reset()
        assert not eng.path_coverage(len(path)).any()
    finally:
        eng.close()
    # through the agent: too far from the path is raised by the very step that gets there
    land = synth.synth_landscape(5, 200, 4)
    line = np.stack([np.linspace(50, 150, 40), np.full(40, 100.0)], axis=1)
    far = navsim_amd.NavBySceneFamiliarity(land, (8, 8), 2.0, n_test_angles=4, max_distance_to_training_path=1.0,
                                           familiarity_model=navsim_amd.sads_familiarity())
    far.train_from_path(line)
    assert far._metrics_on_device
    far.position, far.angle = (100.0, 150.0), 0.0
    with pytest.raises(navsim_amd.TooFarFromTrainingPathException):
        far.step_forward()
    assert far.navigated_for_frames == 1
    far.clear_training()


def test_golden_trajectories_as_ensemble_members(manifest, golden):
    """The reference's golden trajectories (tests/golden/t4_trajectory.npz) with the agent stepping as a member of a
    NavEnsemble: member 0 (the trained agent) and member 1 (a clone at the same pose) must both reproduce the
    reference's headings, positions and angles bit for bit while other members wander elsewhere on the same library --
    sensing and scoring of all of them share the batched passes (dv_sense_step_batch)."""
    z = golden("t4_trajectory.npz")
    for case in manifest["t4_trajectory"]:
        land = synth.synth_landscape(case["landscape"]["seed"], case["landscape"]["size"], case["landscape"]["grain"])
        size = case["landscape"]["size"]
        path = synth.sin_training_path(0.5, 0.2 * size, 0.6 * size, arclen=1.0)[:case["n_views"]]
        nsf = navsim_amd.NavBySceneFamiliarity(
            land, case["sensor_dimensions"], case["step_size"], n_test_angles=case["n_test_angles"],
            sensor_pixel_dimensions=case["sensor_pixel_dimensions"], n_sensor_levels=case["n_sensor_levels"],
            mask_middle_n=case["mask_middle_n"], saccade_degrees=case["saccade_degrees"],
            max_distance_to_training_path=450, familiarity_model=navsim_amd.sads_familiarity(case["chem_weight"]),
            track_scene_familiarity=False)
        nsf.train_from_path(path)
        d = path[2] - path[1]
        a0 = float(np.arctan2(d[1], d[0]) % (2 * np.pi)) + np.deg2rad(case["start_angle_offset_deg"])
        p0 = path[1] + np.array(case["start_offset"])
        others = [(path[k] + np.array([0.5, 0.25 * j]), a0 + 0.1 * j) for j, k in enumerate((30, 90, 150, 200, 250))]
        ens = navsim_amd.NavEnsemble.from_agent(nsf, [(p0, a0), (p0, a0)] + others)
        n = min(case["steps_recorded"], 300)
        name = case["name"]
        for t in range(n):
            ens.step_forward()
            for m in (0, 1):
                ag = ens.agents[m]
                assert ag.last_best_idex == z[name + "_best"][t], (name, m, t)
                assert np.array([ag.position[0], ag.position[1]]).tobytes() == z[name + "_pos"][t].tobytes(), (name, m, t)
                assert np.float64(ag.angle).tobytes() == z[name + "_angle"][t].tobytes(), (name, m, t)
                np.testing.assert_allclose(ag.step_familiarity, z[name + "_fam"][t], rtol=RTOL)
        # the member with the device-side metrics and the clone with the host's: the same numbers
        assert float(ens.agents[0].navigation_error) == float(ens.agents[1].navigation_error)
        assert np.array_equal(ens.agents[0]._coverage_array, ens.agents[1]._coverage_array)
        nsf._engine.close()


def test_one_agent_off_the_landscape_does_not_stop_the_ensemble():
    """The reference's trials are independent: an IndexError in one (its rotated sensor reaches past the landscape,
    util.pyx:137-168) leaves the others running.  In a batched pass the offending agent alone carries
    DV_RES_SENSE_ERROR; NavEnsemble stops it and steps the rest, whose decisions equal the same agents stepped alone."""
    land = synth.synth_landscape(3, 120, 4)
    path = np.stack([np.linspace(40, 80, 30), np.full(30, 60.0)], axis=1)

    def trained():
        a = navsim_amd.NavBySceneFamiliarity(land, (40, 40), 1.0, n_test_angles=4, track_scene_familiarity=False,
                                             familiarity_model=navsim_amd.sads_familiarity(0.25))
        a.train_from_path(path)
        return a

    poses = [((60.0, 60.0), 0.0), ((99.4, 99.4), 0.8 + np.pi / 2), ((50.0, 61.0), 0.2), ((70.0, 58.0), 6.0)]
    alone = []
    for pos, ang in poses:
        a = trained()
        a.position, a.angle = pos, ang
        try:
            a.step_forward(fake=True)
            alone.append((a.last_best_idex, a.position, a.angle))
        except IndexError:
            alone.append("IndexError")
        a._engine.close()
    assert alone[1] == "IndexError" and all(x != "IndexError" for i, x in enumerate(alone) if i != 1)
    ens = navsim_amd.NavEnsemble.from_agent(trained(), poses)
    running = ens.step_forward(fake=True)
    assert running == [0, 2, 3] and ens.stop_status[1] == navsim_amd.NavEnsemble.SENSE_ERROR_STATUS
    assert isinstance(ens.agents[1].stopped_with_exception, IndexError)
    for i in (0, 2, 3):
        ag = ens.agents[i]
        assert (ag.last_best_idex, ag.position, ag.angle) == alone[i], i
    assert ens.step_forward(fake=True) == [0, 2, 3]                   # and the next pass runs without it
    ens.engine.close()


def test_ensemble_share_of_config_five_at_size():
    """One GPU's share of BASELINE.json configs[4]: 32 agents x 16 headings against 100 000 views of 64x64 through
    dv_step_batch (8 library passes of 64 headings).  Planted answers for six agents in different passes, every
    agent's per-heading maxima checked against the oracle on a sample of views that contains its best ones."""
    F, h, w, A, n_agents, seed, cw = 100000, 64, 64, 16, 32, 20261004, 0.25
    eng = navsim_amd.FamiliarityEngine(0)
    try:
        eng.generate_library(seed, F, h, w, cw)
        patches = synth.synth_patches(seed + 5, n_agents * A, h, w).reshape(n_agents, A, h, w, 3)
        planted = {0: (3, 77), 5: (15, 99999), 11: (0, 50000), 17: (8, 31337), 24: (9, 64), 31: (7, 12345)}
        for ag, (a, f) in planted.items():
            v = synth.synth_views(seed, 1, h, w, first_view=f)[0]
            patches[ag, a] = v if ag % 2 else synth.near_match_patch(v, ag + 1, fraction=0.01)
        res = eng.step_batch(patches)
        assert len(res) == n_agents
        for ag, (a, f) in planted.items():
            assert (res[ag]["best_idex"], res[ag]["best_view"]) == (a, f), ag
            if ag % 2:
                assert res[ag]["step_familiarity"] == float(h * w)
        # sampled views: the reported best view of every (agent, heading) plus a fixed spread; on that sample the
        # oracle's per-heading maximum must be the reported one (it contains the maximiser) and no sampled view may beat it
        spread = np.arange(0, F, 4999)
        for ag in range(n_agents):
            views = np.unique(np.concatenate([spread, np.asarray(res[ag]["angle_view"], dtype=np.int64)]))
            lib = np.stack([synth.synth_views(seed, 1, h, w, first_view=int(f))[0] for f in views])
            want = oracle.step(lib, patches[ag], cw)
            np.testing.assert_allclose(res[ag]["angle_familiarity"], want["angle_familiarity"], rtol=RTOL)
            assert res[ag]["best_idex"] == want["best_idex"], ag
            assert res[ag]["best_view"] == int(views[want["best_view"]]), ag
    finally:
        eng.close()


def test_ssd_f32_full_size_properties():
    """ssd_f32 on BASELINE configs[1]'s shape (64x64, 50 000 views; 16 and 32 headings = one and two passes): planted
    near-copies win at their headings, the planted pairs' SSDs equal the reference's `ssds` on the upcast data to 1e-6,
    scaling every input by 2 scales every SSD by exactly 4, and both workgroup shapes agree on the decision."""
    import os
    F, h, w = 50000, 64, 64
    rng = np.random.default_rng(2)
    lib = rng.random((F, h, w), dtype=np.float32)
    for A in (16, 32):
        patches = rng.random((A, h, w), dtype=np.float32)
        patches[A - 3] = lib[31337] + np.float32(0.01)
        patches[2] = lib[49999] + rng.normal(0, 0.02, (h, w)).astype(np.float32)
        seen = {}
        for shape in ("0", "2"):                    # default (single wave, prefetch) and four waves + LDS fold
            os.environ["DEJAVU_SHAPE"] = shape
            try:
                eng = navsim_amd.FamiliarityEngine(0)
            finally:
                os.environ.pop("DEJAVU_SHAPE", None)
            try:
                eng.set_library_f32(lib)
                r = eng.step_f32(patches)
                assert (r["best_idex"], r["best_view"]) == (A - 3, 31337)
                assert int(r["angle_view"][2]) == 49999
                for a, f in ((A - 3, 31337), (2, 49999)):
                    want = oracle.ssds(patches[a].astype(np.float64), lib[f].astype(np.float64))
                    np.testing.assert_allclose(r["angle_ssd"][a], want, rtol=1e-6)
                seen[shape] = np.array(r["angle_ssd"])
                if shape == "0" and A == 16:
                    eng.set_library_f32(lib * np.float32(2))
                    r2 = eng.step_f32(patches * np.float32(2))
                    assert (r2["best_idex"], r2["best_view"]) == (A - 3, 31337)
                    assert np.array_equal(np.array(r2["angle_ssd"]), 4.0 * seen["0"])      # powers of two: exact
            finally:
                eng.close()
        np.testing.assert_allclose(seen["0"], seen["2"], rtol=1e-6)


def test_experiment_rows_match_the_reference_on_gpu(manifest, golden, tmp_path):
    """One trial of the reference's farm through the product agent (everything on the GPU): the result row and the CSV
    text the reference's run_experiment + row formatting produced (tests/golden manifest "t7_experiment")."""
    from navsim_amd import experiment
    t7 = manifest["t7_experiment"]
    land = synth.synth_landscape(t7["landscape"]["seed"], t7["landscape"]["size"], t7["landscape"]["grain"])
    size = t7["landscape"]["size"]
    tp = synth.sin_training_path(0.5, 0.2 * size, 0.6 * size, arclen=1.0)[:t7["n_views"]]
    done = []
    for row in t7["rows"]:
        nsf = navsim_amd.NavBySceneFamiliarity(land, (16, 8), 1.0, n_test_angles=10, sensor_pixel_dimensions=[2, 4],
                                               n_sensor_levels=4, mask_middle_n=1, saccade_degrees=90.0,
                                               max_distance_to_training_path=450,
                                               familiarity_model=navsim_amd.sads_familiarity(row["chem_weight"]))
        nsf.train_from_path(tp)
        d = tp[2] - tp[1]
        nsf.angle = float(np.arctan2(d[1], d[0]) % (2 * np.pi)) + np.deg2rad(7.0)
        nsf.position = tp[1] + np.array([1.5, -1.0])
        res = experiment.run_experiment(nsf, frames=row["frames"])
        assert experiment.csv_row(row["trial"], res) == row["line"], row["name"]
        assert res["stop_status"] == row["result"]["stop_status"] and float(res["rmsd_error"]) == row["result"]["rmsd_error"]
        done.append((row["trial"], res))
        nsf._engine.close()
    out = tmp_path / "task-0.csv"
    experiment.write_task_csv(str(out), done)
    assert out.read_text().splitlines() == [t7["rows"][0]["header"]] + [r["line"] for r in t7["rows"]]


@pytest.mark.parametrize("cw", [0.0, 0.25])
def test_appended_library_equals_one_ingest(eng, cw):
    """dv_append_library: views appended behind a resident library (only the new view groups are re-tiled, also when
    the old library ends inside a group) give the stored planes and the decisions of one ingest of all of them; views
    the resident layout cannot hold are refused."""
    F0, F1, h, w, A = 700, 1011, 12, 10, 9
    lib = synth.synth_views(41, F1, h, w)
    pats = synth.synth_patches(41, A, h, w)
    pats[4] = synth.near_match_patch(lib[903], 2)             # best view among the appended ones
    eng.set_library(lib, cw)
    want_planes = eng.read_planes(0, F1)
    want = eng.step(pats, want_scene=True)
    eng.set_library(lib[:F0], cw)
    eng.step(pats, want_scene=False)                          # (times the kernel forms on the old library)
    eng.append_library(lib[F0:900])
    eng.append_library(lib[900:])
    info = eng.library_info()
    assert info["n_views"] == F1
    assert np.array_equal(eng.read_planes(0, F1), want_planes)
    got = eng.step(pats, want_scene=True)
    assert (got["best_idex"], got["best_view"]) == (want["best_idex"], want["best_view"]) == (4, 903)
    assert np.array_equal(got["angle_familiarity"], want["angle_familiarity"])
    assert np.array_equal(got["scene_familiarity"], want["scene_familiarity"])
    ref = oracle.step(lib, pats, cw)
    assert (got["best_idex"], got["best_view"]) == (ref["best_idex"], ref["best_view"])
    if cw > 0:
        odd = lib[:3].copy()
        odd[..., 0] = 33                                       # a hue the resident layout has no plane for
        odd[..., 1] = 50
        with pytest.raises(navsim_amd.EngineError):
            eng.append_library(odd)
        assert eng.library_info()["n_views"] == F1             # refused before anything changed


def test_agent_trains_on_a_second_path():
    """train_additional_path: the library, the path and the metrics grow; the agent with everything on the GPU (views
    of the second path sensed and appended on the device) walks exactly like the host agent over the oracle, and covers
    points of the second path."""
    land = synth.synth_landscape(11, 420, 4)
    p1 = synth.sin_training_path(0.5, 0.2 * 420, 0.6 * 420, arclen=1.0)[:150]
    p2 = np.stack([np.linspace(120, 300, 181), np.full(181, 90.0)], axis=1)
    host = navsim_amd.NavBySceneFamiliarity(land, (16, 12), 1.0, n_test_angles=7, use_gpu_sensor=False,
                                            familiarity_model=oracle.sads_familiarity(0.25))
    nsf = navsim_amd.NavBySceneFamiliarity(land, (16, 12), 1.0, n_test_angles=7, track_scene_familiarity=False,
                                           familiarity_model=navsim_amd.sads_familiarity(0.25))
    for a in (host, nsf):
        a.train_from_path(p1)
        a.train_additional_path(p2)
        a.position, a.angle = (p2[5][0] + 0.5, p2[5][1] - 0.5), 0.05
    assert np.array_equal(nsf.familiar_scenes, host.familiar_scenes) and len(nsf.training_path) == 331
    assert nsf.training_path_length == host.training_path_length
    for _ in range(60):
        host.step_forward()
        nsf.step_forward()
        assert nsf.last_best_idex == host.last_best_idex and nsf.position == host.position
    assert float(nsf.navigation_error) == float(host.navigation_error)
    assert np.array_equal(nsf._coverage_array, host._coverage_array) and nsf._coverage_array[150:].any()
    nsf._engine.close()


@pytest.mark.parametrize("finish", ["2", "0"])
def test_fenced_and_unfenced_arrival_tickets_decide_alike(finish):
    """k_finish / k_tail hand their per-block results to the last block through agent-scope atomics; the release /
    acquire pair of the memory model around the arrival ticket is optional on the single-agent integer path
    (DEJAVU_FENCED, csrc/dejavu_hip.hip:step_fenced).  Both forms must give the reference's decision, per-heading maxima
    and per-view minima on a library whose best view is duplicated in far-apart blocks and seen by several headings."""
    import os
    F, h, w, A, cw = 90000, 4, 4, 16, 0.25
    lib = synth.synth_views(23, F, h, w)
    star = lib[41000].copy()
    star[0, 0] = (77, 200, 13)
    for f in (5, 30000, 41000, 89999):
        lib[f] = star
    pats = synth.synth_patches(23, A, h, w)
    pats[3] = pats[9] = pats[15] = star
    want = oracle.step(lib, pats, cw)
    seen = {}
    for fenced in ("0", "1"):
        os.environ["DEJAVU_FENCED"], os.environ["DEJAVU_FINISH"] = fenced, finish
        try:
            e = navsim_amd.FamiliarityEngine(0)
        finally:
            os.environ.pop("DEJAVU_FENCED", None)
            os.environ.pop("DEJAVU_FINISH", None)
        try:
            e.set_library(lib, cw)
            for _ in range(20):
                r = e.step(pats, want_scene=True)
                assert (r["best_idex"], r["best_view"]) == (want["best_idex"], want["best_view"]) == (3, 5), fenced
            seen[fenced] = (np.array(r["angle_familiarity"]), np.array(r["scene_familiarity"]), r["n_candidates"])
        finally:
            e.close()
    assert np.array_equal(seen["0"][0], seen["1"][0]) and np.array_equal(seen["0"][1], seen["1"][1])
    assert seen["0"][2] == seen["1"][2] >= 12
    np.testing.assert_allclose(seen["0"][0], want["angle_familiarity"], rtol=RTOL)


@pytest.mark.parametrize("fp4,tiles", [("1", "1"), ("0", "1"), ("1", "2"), ("0", "2")])
def test_matrix_core_kernel_forms_under_repetition(fp4, tiles):
    """k_sad_mfma_dual streams both operands through LDS rings with hand-counted waits (fp4 and int8 bodies, one or two
    view groups per wave).  A race would show as an occasional wrong sum: 150 steps with changing patches on a library
    with ragged view-group ranges (waves without a group of their own in most items), every decision, per-heading
    maximum and per-view minimum against the oracle (the integer sums are exact, so the scores must agree to the last
    bit from step to step)."""
    import os
    F, h, w, A, cw = 9000 + 37, 20, 12, 13, 0.25
    lib = synth.synth_views(61, F, h, w)
    os.environ.update(DEJAVU_SHAPE="6", DEJAVU_BITS="2", DEJAVU_FP4=fp4, DEJAVU_MFMA_TILES=tiles)
    try:
        e = navsim_amd.FamiliarityEngine(0)
    finally:
        for k in ("DEJAVU_SHAPE", "DEJAVU_BITS", "DEJAVU_FP4", "DEJAVU_MFMA_TILES"):
            os.environ.pop(k, None)
    try:
        e.set_library(lib, cw)
        assert e.library_info()["has_bit_planes"]
        for it in range(150):
            pats = synth.synth_patches(1000 + it % 5, A, h, w)
            pats[it % A] = synth.near_match_patch(lib[(it * 997) % F], it, fraction=0.02)
            got = e.step(pats, want_scene=True)
            fused = e.step(pats, want_scene=False)                   # the step as the agent runs it: finished in the kernel's epilogue
            assert np.array_equal(fused["angle_familiarity"], got["angle_familiarity"]), it
            assert np.array_equal(fused["angle_view"], got["angle_view"]) and fused["best_idex"] == got["best_idex"], it
            if it < 15:
                want = oracle.step(lib, pats, cw)
                assert (got["best_idex"], got["best_view"]) == (want["best_idex"], want["best_view"]), it
                np.testing.assert_allclose(got["angle_familiarity"], want["angle_familiarity"], rtol=RTOL)
                np.testing.assert_allclose(got["scene_familiarity"], want["scene_familiarity"], rtol=RTOL)
            assert got["best_view"] == (it * 997) % F and got["best_idex"] == it % A, it
    finally:
        e.close()


def test_ties_inside_one_finishing_block_of_several_view_sets():
    """On large libraries a k_finish block owns several consecutive sets of 256 views and keeps ONE representative per
    heading; a set's representative that loses to a later set's must still reach the candidate list.  The best view is
    duplicated in two sets of the same block (and far away), seen by two headings: the decision, every per-heading
    maximum and the candidate count must be the oracle's / the two-kernel ending's."""
    import os
    F, h, w, A, cw = 300000, 4, 4, 20, 0.25                     # 300 000 views -> 3 view sets per block
    lib = synth.synth_views(29, F, h, w)
    star = lib[100000].copy()
    star[0, 0] = (77, 200, 13)
    dup = (2 * 768 + 700, 2 * 768 + 10, 2 * 768 + 300, 200000, 299999)       # three in block 2 (sets 2, 0, 1), two elsewhere
    for f in dup:
        lib[f] = star
    pats = synth.synth_patches(29, A, h, w)
    pats[7] = pats[18] = star
    want = oracle.step(lib, pats, cw)
    assert (want["best_idex"], want["best_view"]) == (7, min(dup))
    seen = {}
    for finish in ("2", "0"):
        os.environ["DEJAVU_FINISH"] = finish
        try:
            e = navsim_amd.FamiliarityEngine(0)
        finally:
            os.environ.pop("DEJAVU_FINISH", None)
        try:
            e.set_library(lib, cw)
            r = e.step(pats, want_scene=True)
            assert (r["best_idex"], r["best_view"]) == (want["best_idex"], want["best_view"]), finish
            np.testing.assert_allclose(r["angle_familiarity"], want["angle_familiarity"], rtol=RTOL)
            np.testing.assert_allclose(r["scene_familiarity"], want["scene_familiarity"], rtol=RTOL)
            assert r["flags"] & 1                                   # resolved exactly
            seen[finish] = r["n_candidates"]
        finally:
            e.close()
    assert seen["2"] == seen["0"] >= 2 * len(dup)


def _engine_with(env):
    import os
    keys = ("DEJAVU_SHAPE", "DEJAVU_BITS", "DEJAVU_FP4", "DEJAVU_FUSE", "DEJAVU_MFMA_TILES", "DEJAVU_VCODE")
    before = {k: os.environ.pop(k, None) for k in keys}
    os.environ.update(env)
    try:
        return navsim_amd.FamiliarityEngine(0)
    finally:
        for k in keys:
            os.environ.pop(k, None)
            if before[k] is not None:
                os.environ[k] = before[k]


@pytest.mark.parametrize("F,h,w,A,cw,tiles", [(5000, 32, 32, 32, 0.5, "0"), (3001, 20, 24, 7, 0.3, "2"), (9037, 16, 16, 64, 0.5, "0"),
                                               (4100, 32, 32, 16, 0.0, "2"), (2500, 12, 20, 33, 1.0, "0"),
                                               (70001, 16, 16, 32, 0.25, "0"), (41000, 8, 24, 9, 0.0, "0")])
def test_fp4_form_gives_the_int8_forms_sums(F, h, w, A, cw, tiles):
    """The fp4 form of the matrix-core kernel (on-level patches, one gap width per nibble bit) must leave the very
    integer sums of the int8 form: scores, per-heading maxima and decisions identical to the last bit, with the
    finishing epilogue inside the kernel or outside it; and both agree with the oracle."""
    lib = synth.synth_views(11, F, h, w)
    pat = synth.synth_patches(11, A, h, w)            # the library's own level set
    pat[A // 2] = synth.near_match_patch(lib[F // 3], 5, fraction=0.03)
    want = oracle.step(lib, pat, cw)
    seen = {}
    for fp4 in ("1", "code", "0"):                    # "code": the value plane as 3-bit level codes (DEJAVU_VCODE=1, k_bitpack_code)
        for fuse in ("1", "0"):
            e = _engine_with(dict(DEJAVU_SHAPE="6", DEJAVU_BITS="2", DEJAVU_FP4="0" if fp4 == "0" else "1", DEJAVU_FUSE=fuse,
                                  DEJAVU_MFMA_TILES=tiles, DEJAVU_VCODE="1" if fp4 == "code" else "0"))
            try:
                e.set_library(lib, cw)
                info = e.library_info()
                assert info["has_bit_planes"] and info["fp4_form"] == (fp4 != "0")
                assert (info["code_tile_bytes"] > 0) == (fp4 == "code" and cw < 1.0)
                if info["code_tile_bytes"]:
                    assert info["code_tile_bytes"] < info["bit_tile_bytes"]
                got = e.step(pat, want_scene=False)
                if fp4 != "0":
                    assert e.patches_on_level()
                assert (got["best_idex"], got["best_view"]) == (want["best_idex"], want["best_view"]) == (A // 2, F // 3)
                np.testing.assert_allclose(got["angle_familiarity"], want["angle_familiarity"], rtol=RTOL, atol=1e-12)
                scene = e.step(pat, want_scene=True)["scene_familiarity"]
                np.testing.assert_allclose(scene, want["scene_familiarity"], rtol=RTOL, atol=1e-12)
                fam = np.empty(F)
                e.score(pat[0], fam)
                seen[fp4, fuse] = (np.array(got["angle_familiarity"]), np.array(got["angle_view"]), scene.copy(), fam, got["n_candidates"])
            finally:
                e.close()
    first = seen["1", "1"]
    for key, other in seen.items():
        for x, y in zip(first[:4], other[:4]):
            assert np.array_equal(x, y), key
        assert first[4] == other[4], key


def test_off_level_patches_take_the_int8_form_in_the_same_launch():
    """A patch byte strictly inside a gap of the library's levels has no fp4 coefficient: k_patch_prep flags the prep and
    the same launch scores with the int8 image.  Alternating on-level and off-level patch sets on one engine: every
    step against the oracle, and the flag follows the patches."""
    F, h, w, A, cw = 6000 + 5, 24, 16, 12, 0.4
    lib = synth.synth_views(23, F, h, w)
    e = _engine_with(dict(DEJAVU_SHAPE="6", DEJAVU_BITS="2"))
    try:
        e.set_library(lib, cw)
        assert e.library_info()["fp4_form"]
        for it in range(8):
            pat = synth.synth_patches(100 + it, A, h, w)
            off = it % 2 == 1
            if off:
                pat[it % A, it % h, (3 * it) % w, 2] ^= 0x10          # one V byte off its level
            pat[(it + 1) % A] = synth.near_match_patch(lib[(it * 811) % F], it, fraction=0.02)
            want = oracle.step(lib, pat, cw)
            got = e.step(pat, want_scene=True)
            assert e.patches_on_level() == (not off), it
            assert (got["best_idex"], got["best_view"]) == (want["best_idex"], want["best_view"]), it
            np.testing.assert_allclose(got["angle_familiarity"], want["angle_familiarity"], rtol=RTOL, atol=1e-12)
            np.testing.assert_allclose(got["scene_familiarity"], want["scene_familiarity"], rtol=RTOL, atol=1e-12)
    finally:
        e.close()


def test_fp4_form_needs_one_gap_width_per_nibble_bit():
    """K-element n = plane n % T sits on bit n % 4 of a nibble.  Five V levels (four planes) put ONE plane on each bit, so
    the widths 63, 64, 64, 64 of the reference's five-level quantiser are fine; four levels with unequal gaps (three
    planes: every plane lands on every bit) are not, and keep the int8 form; equal gaps are fine again, and so is a gap wider
    than 127 that the int8 form had to split into copies."""
    F, h, w = 700, 8, 8
    rng = np.random.default_rng(3)
    for levels, ok in (([0, 63, 127, 191, 255], True), ([0, 60, 130, 255], False), ([0, 85, 170, 255], True), ([0, 255], True), ([0, 100], True),
                       ([0, 127, 255], False), ([3, 200], True)):
        lib = np.zeros((F, h, w, 3), np.uint8)
        lib[..., 2] = np.array(levels, np.uint8)[rng.integers(0, len(levels), (F, h, w))]
        pat = np.zeros((3, h, w, 3), np.uint8)
        pat[..., 2] = np.array(levels, np.uint8)[rng.integers(0, len(levels), (3, h, w))]
        e = _engine_with(dict(DEJAVU_SHAPE="6", DEJAVU_BITS="2"))
        try:
            e.set_library(lib, 0.0)
            info = e.library_info()
            # (0, 255: one gap split 127 + 127 + 1 for int8 -- its first plane stands for all 255 in the fp4 form; 0, 127, 255:
            #  widths 127 and 128 meet on every bit position)
            assert info["has_bit_planes"] and info["fp4_form"] == ok, levels
            got = e.step(pat, want_scene=True)
            want = oracle.step(lib, pat, 0.0)
            assert (got["best_idex"], got["best_view"]) == (want["best_idex"], want["best_view"]), levels
            np.testing.assert_allclose(got["angle_familiarity"], want["angle_familiarity"], rtol=RTOL, atol=1e-12)
            np.testing.assert_allclose(got["scene_familiarity"], want["scene_familiarity"], rtol=RTOL, atol=1e-12)
        finally:
            e.close()


@pytest.mark.parametrize("A", [32, 12])
def test_two_level_fold_with_ties_across_slices(eng, A):
    """More than 512 summaries per agent: k_fold_reduce cuts them to 32 on as many workgroups before k_fold decides.  The
    best view duplicated in slices far apart (and twice inside one slice), seen under three headings; the oracle's
    decision (first heading, first view), every per-heading maximum, and the candidate count."""
    F, h, w, cw = 300000 + 123, 4, 4, 0.25
    lib = synth.synth_views(19, F, h, w)
    pats = synth.synth_patches(19, A, h, w)
    star = lib[150000].copy()
    star[0, 0] = (77, 200, 13)                          # unlike the 2-hue, 5-level crowd: a unique best match
    dup = (299999, 150000, 150300, 801, 77000)
    for f in dup:
        lib[f] = star
    for a in (A - 1, A // 2, 2):
        pats[a] = star
    want = oracle.step(lib, pats, cw)
    assert (want["best_idex"], want["best_view"]) == (2, 801)
    eng.set_library(lib, cw)
    got = eng.step(pats, want_scene=False)
    assert (got["best_idex"], got["best_view"]) == (2, 801)
    assert got["n_candidates"] == 3 * len(dup) and got["flags"] & 1
    np.testing.assert_allclose(got["angle_familiarity"], want["angle_familiarity"], rtol=1e-12)
    assert list(got["angle_view"]) == list(want["angle_view"]) if "angle_view" in want else True
