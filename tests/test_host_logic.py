"""CPU tests of the host side: sensor model, agent loop, synthetic generator, C-ABI surface.

The agent is driven here with a TEST-ONLY familiarity plug-in backed by the oracle (the reference's
own plug-in point, NavBySceneFamiliarity.py:72), so that the host logic around the GPU call --
sensor extraction, heading update, error metrics, stop conditions -- is pinned against the
reference's golden trajectories without a GPU.  The product default (HIP engine) is tested under
-m gpu.
"""
import ctypes
import os
import re
import sys

import numpy as np
import pytest

import navsim_amd
from navsim_amd import synth
from navsim_amd import agent as agent_mod
from oracle import oracle
from tests.helpers import sha
from tests.conftest import REPO


def test_synth_is_deterministic():
    v = synth.synth_views(3, 5, 4, 6, first_view=7)
    assert v.shape == (5, 4, 6, 3) and v.dtype == np.uint8
    # view f depends only on (seed, first_view + f)
    w = synth.synth_views(3, 2, 4, 6, first_view=9)
    assert np.array_equal(v[2:4], w)
    assert set(np.unique(v[..., 2])) <= {0, 63, 127, 191, 255}
    assert set(np.unique(v[..., 0])) <= {0, 127} and set(np.unique(v[..., 1])) <= {0, 127}
    assert sha(synth.synth_views(1, 3, 2, 2)) == sha(synth.synth_views(1, 3, 2, 2))
    assert int(synth.splitmix64(np.array([0], dtype=np.uint64))[0]) == 0
    assert int(synth.splitmix64(np.array([1], dtype=np.uint64))[0]) == 0x5692161D100B05E5


def test_sensor_model_matches_reference(manifest, golden):
    z = golden("t5_sensor.npz")
    meta = manifest["t5_sensor"]["meta"]
    land = synth.synth_landscape(meta["landscape"]["seed"], meta["landscape"]["size"], meta["landscape"]["grain"])
    assert sha(land) == meta["landscape"]["sha"]
    lands = {"land": land, "land2": z["land2"]}
    assert sha(lands["land2"]) == meta["land2_sha"]
    for case in manifest["t5_sensor"]["cases"]:
        levels = case["n_sensor_levels"]
        levels = tuple(levels) if isinstance(levels, list) else levels
        nsf = navsim_amd.NavBySceneFamiliarity(
            lands[case["landscape"]], case["sensor_dimensions"], 1.0, n_test_angles=4,
            sensor_pixel_dimensions=case["sensor_pixel_dimensions"], n_sensor_levels=levels,
            mask_middle_n=case["mask_middle_n"], familiarity_model=oracle.sads_familiarity())
        for k, (x, y, a) in enumerate(case["poses"]):
            mat = nsf.get_sensor_mat((x, y), a)
            assert np.array_equal(nsf._landscape_glimpse_buf, z[case["name"] + "_glimpses"][k]), (case["name"], k)
            assert np.array_equal(mat, z[case["name"] + "_mats"][k]), (case["name"], k)
    # integer-division quirk of the saturation average (util.pyx:131): 8x2 block of S=255 -> 252
    assert np.array_equal(agent_mod.downscale_chem(z["dc_quirk_in"], 8, 2), z["dc_quirk_out"])


def _run_trajectory(case, land, model, **agent_kwargs):
    path = synth.sin_training_path(0.5, 0.2 * case["landscape"]["size"], 0.6 * case["landscape"]["size"],
                                   arclen=1.0)[:case["n_views"]]
    assert sha(path) == case["path_sha"]
    nsf = navsim_amd.NavBySceneFamiliarity(
        land, case["sensor_dimensions"], case["step_size"], n_test_angles=case["n_test_angles"],
        sensor_pixel_dimensions=case["sensor_pixel_dimensions"], n_sensor_levels=case["n_sensor_levels"],
        mask_middle_n=case["mask_middle_n"], saccade_degrees=case["saccade_degrees"],
        max_distance_to_training_path=450, familiarity_model=model, **agent_kwargs)
    nsf.train_from_path(path)
    d = path[2] - path[1]
    nsf.angle = float(np.arctan2(d[1], d[0]) % (2 * np.pi)) + np.deg2rad(case["start_angle_offset_deg"])
    nsf.position = path[1] + np.array(case["start_offset"])
    best, pos, ang, fam = [], [], [], []
    status = 0
    try:
        for _ in range(case["n_steps"]):
            nsf.step_forward()
            best.append(nsf.last_best_idex)
            pos.append([nsf.position[0], nsf.position[1]])
            ang.append(nsf.angle)
            fam.append(nsf.step_familiarity)
    except navsim_amd.StopNavigationException as e:
        status = e.get_code()
    return nsf, np.array(best), np.array(pos), np.array(ang), np.array(fam), status


def check_trajectory(case, z, model, fam_rtol, **agent_kwargs):
    land = synth.synth_landscape(case["landscape"]["seed"], case["landscape"]["size"], case["landscape"]["grain"])
    assert sha(land) == case["landscape"]["sha"]
    nsf, best, pos, ang, fam, status = _run_trajectory(case, land, model, **agent_kwargs)
    name = case["name"]
    assert sha(nsf.familiar_scenes) == bytes(z[name + "_scenes_sha"]).hex()      # training views byte-identical
    n = case["steps_recorded"]
    assert status == case["stop_status"] and len(best) == n
    assert np.array_equal(best, z[name + "_best"])                   # heading index: bit-identical
    assert pos.tobytes() == z[name + "_pos"].tobytes()               # hence the same trajectory, bit for bit
    assert ang.tobytes() == z[name + "_angle"].tobytes()
    np.testing.assert_allclose(fam, z[name + "_fam"], rtol=fam_rtol, atol=0)
    np.testing.assert_allclose(nsf.angle_familiarity, z[name + "_last_angle_fam"], rtol=fam_rtol)
    np.testing.assert_allclose(nsf.scene_familiarity, z[name + "_last_scene_fam"], rtol=fam_rtol)
    assert nsf.navigated_for_frames == case["navigated_for_frames"]
    assert float(nsf.navigation_error) == case["navigation_error"]
    assert float(nsf.percent_recapitulated) == case["percent_recapitulated"]
    assert float(nsf.percent_recapitulated_forgiving(0.05)) == case["percent_forgiving"]
    assert int(nsf.n_captures(0.05)) == case["n_captures"]
    assert float(nsf.training_path_length) == case["training_path_length"]


def test_agent_trajectories_match_reference_with_oracle_plugin(manifest, golden):
    z = golden("t4_trajectory.npz")
    for case in manifest["t4_trajectory"]:
        check_trajectory(case, z, oracle.sads_familiarity(case["chem_weight"]), fam_rtol=0)


def test_agent_api_and_stop_conditions():
    land = synth.synth_landscape(5, 200, 4)
    nsf = navsim_amd.NavBySceneFamiliarity(land, (8, 8), 2.0, n_test_angles=4,
                                           familiarity_model=oracle.sads_familiarity())
    for attr in ("position", "angle", "angle_offsets", "angle_familiarity", "scene_familiarity", "step_familiarity",
                 "familiar_scenes", "training_path", "training_path_length", "navigated_for_frames",
                 "stopped_with_exception", "n_test_angles", "step_size", "sensor_dimensions",
                 "sensor_pixel_dimensions", "n_sensor_levels", "mask_middle_n", "saccade_degrees"):
        assert hasattr(nsf, attr), attr
    assert nsf.n_sensor_levels == (256, 256, 5)
    assert np.allclose(nsf.angle_offsets, np.linspace(-np.pi / 2, np.pi / 2, 4))
    path = np.stack([np.linspace(50, 150, 40), np.full(40, 100.0)], axis=1)
    nsf.train_from_path(path)
    with pytest.raises(ValueError):
        nsf.train_from_path(path)                       # one-shot (NavBySceneFamiliarity.py:119-120)
    assert nsf.familiar_scenes.shape == (40, 8, 8, 3) and nsf.scene_familiarity.shape == (40,)
    # walks to the end of the path and says so
    nsf.position, nsf.angle = path[-1] - np.array([3.0, 0.0]), 0.0
    with pytest.raises(navsim_amd.ReachedEndOfTrainingPathException) as ei:
        nsf.step_forward()                              # ends within threshold_factor * step_size (:328)
    assert ei.value.get_code() == 1 and str(ei.value) == "agent reached end of training path"
    # out of bounds is raised before anything is scored
    nsf.position = (2.0, 100.0)
    with pytest.raises(navsim_amd.OutOfLandscapeBoundsException) as ei:
        nsf.step_forward()
    assert ei.value.get_code() == -2 and np.all(np.isnan(nsf.angle_familiarity))
    # too far from the training path
    nsf.clear_training()
    far = navsim_amd.NavBySceneFamiliarity(land, (8, 8), 2.0, n_test_angles=4, max_distance_to_training_path=1.0,
                                           familiarity_model=oracle.sads_familiarity())
    far.train_from_path(path)
    far.position, far.angle = (100.0, 150.0), 0.0
    with pytest.raises(navsim_amd.TooFarFromTrainingPathException) as ei:
        far.step_forward()
    assert ei.value.get_code() == -1
    assert issubclass(navsim_amd.TooFarFromTrainingPathException, navsim_amd.NavigatingFailedException)
    assert issubclass(navsim_amd.NavigatingFailedException, navsim_amd.StopNavigationException)


def test_c_abi_exports_every_declared_symbol():
    """The library loads and exports exactly what include/dejavu.h declares (no compute calls here)."""
    from navsim_amd import _native
    header = open(os.path.join(REPO, "include", "dejavu.h")).read()
    declared = set(re.findall(r"\b(dv_[a-z0-9_]+)\s*\(", header))
    assert declared == set(_native.PROTOTYPES), declared ^ set(_native.PROTOTYPES)
    lib = _native.load()
    for name in declared:
        assert hasattr(lib, name), name
    assert b"gfx950" in lib.dv_version()
    # struct layout agreed between header and binding
    assert ctypes.sizeof(_native.StepResult) == 56 + 4 * 8 * 64
    # dv_last_error(NULL) is safe without a context
    assert isinstance(lib.dv_last_error(None), bytes)


def test_shipped_scoring_kernels_use_no_scratch():
    """No kernel of the built library keeps registers in scratch memory (read from the code object inside libdejavu_hip.so:
    tools/kernel_resources.py) -- the matrix-core loop's hand-counted waits (inline-asm ds_read_b128 / LDS-DMA) are only right
    while the compiler neither copies nor spills the registers involved.  Two instantiations are known to spill a few
    item-level values and are never launched by default: the round-2 body with two view groups per wave in its fused form
    (launch_mfma asks hipFuncGetAttributes and takes one view group per wave while it spills) and the two-heading-tile body on
    3-bit code rows (DEJAVU_VCODE=1, an experiment's knob)."""
    sys.path.insert(0, os.path.join(REPO, "tools"))
    try:
        import kernel_resources
    finally:
        sys.path.pop(0)
    rows = kernel_resources.kernel_table()
    assert len(rows) > 100 and any(r["name"].startswith("k_sad_mfma_dual<") for r in rows)
    guarded = {"k_sad_mfma_dual<1, 3, 2, 3, 2, true, 0, 3, false, 1>", "k_sad_mfma_dual<4, 2, 2, 4, 1, true, 4, 3, true, 2>"}
    spilling = {r["name"]: r["scratch"] for r in rows if r.get("vgpr_spills", 0) > 0}
    assert set(spilling) <= guarded, "kernels with spilled registers: %r" % (spilling,)
    # scratch that is not a spill: the two exact fp64 kernels index a small private array of plane bytes (96 bytes per lane)
    private = {r["name"] for r in rows if r["scratch"] > 0 and r.get("vgpr_spills", 0) == 0}
    assert private <= {"k_exact_all", "k_resolve"}, private
    for r in rows:                                                    # every step's fold, whatever its thread count
        if r["name"].startswith("k_fold<"):
            assert r["scratch"] == 0, r
    # the loader / consumer body that ships (fused and unfused, one and two heading tiles) and the SSD matrix-core kernels
    for name in ("k_sad_mfma_dual<4, 2, 2, 4, 1, true, 4, 3, false, 1>", "k_sad_mfma_dual<4, 2, 2, 4, 1, false, 4, 3, false, 1>",
                 "k_sad_mfma_dual<4, 2, 2, 4, 1, true, 4, 3, false, 2>", "k_sad_mfma_dual<4, 2, 2, 4, 1, false, 4, 3, false, 2>",
                 "k_sad_lc22<2, 3>", "k_ssd_u8_mfma<1>", "k_ssd_u8_mfma<2>"):
        row = [r for r in rows if r["name"] == name]
        assert row and row[0]["scratch"] == 0 and row[0]["vgpr"] <= 256, name


def test_ssd_plugin_factory_arguments():
    """ssd_familiarity(channel) checks its argument like the reference's factory checks chem_weight (util.pyx:12); the two
    later stages need the GPU (tests/test_gpu_parity.py)."""
    with pytest.raises(ValueError):
        navsim_amd.ssd_familiarity(channel=3)
    model = navsim_amd.ssd_familiarity(channel=1)
    assert model.metric == "ssd" and model.channel == 1 and callable(model.make_engine) and callable(model.from_engine)


def test_no_cpu_fallback_in_product(monkeypatch):
    """Scoring must fail loudly when the HIP library is missing."""
    from navsim_amd import _native
    monkeypatch.setattr(_native, "_lib", None)
    monkeypatch.setattr(_native, "LIB_PATH", "/nonexistent/libdejavu_hip.so")
    with pytest.raises(navsim_amd.EngineError):
        navsim_amd.FamiliarityEngine()
    src = ""
    pkg = os.path.join(REPO, "navigation-by-deja-vu_amd", "navsim_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            src += open(os.path.join(pkg, fn)).read()
    assert "import oracle" not in src and "from oracle" not in src


def test_experiment_loop_reports_the_reference_row():
    """run_experiment / chop_path_to_len (scripts/run_experiment.py:107-124,235-258)."""
    land = synth.synth_landscape(5, 200, 4)
    path = np.stack([np.linspace(40, 160, 61), np.full(61, 100.0)], axis=1)      # 2 px apart
    chopped = navsim_amd.chop_path_to_len(path, 100.0)
    seg = np.linalg.norm(chopped[1:] - chopped[:-1], axis=1)
    assert np.sum(seg) <= 100.0 < np.sum(seg) + 4.0 + 1e-9
    assert abs((chopped[0, 0] - path[0, 0]) - (path[-1, 0] - chopped[-1, 0])) <= 2.0 + 1e-9     # trimmed from both ends
    nsf = navsim_amd.NavBySceneFamiliarity(land, (8, 8), 2.0, n_test_angles=5,
                                           familiarity_model=oracle.sads_familiarity())
    nsf.train_from_path(chopped)
    nsf.position, nsf.angle = chopped[1], 0.0
    row = navsim_amd.run_experiment(nsf)
    assert set(row) == {"path_coverage", "rmsd_error", "completed_frames", "stop_status", "percent_forgiving", "n_captures"}
    assert row["stop_status"] in (0, 1, -1, -2)
    assert row["completed_frames"] <= int(3.0 * nsf.training_path_length / nsf.step_size)
    if row["stop_status"] == 1:
        assert isinstance(nsf.stopped_with_exception, navsim_amd.ReachedEndOfTrainingPathException)
        assert row["path_coverage"] > 0.5


def test_bit_plane_decomposition_is_exact():
    """The identity behind the matrix-core scoring path (csrc/dejavu_kernels.h, 'bit-plane library'): for a library
    byte b from the level set and ANY patch byte a,
        |a - b| = (lmin - a)+ + (a - lmax)+ + sum_t alpha_t + sum_t bit_t(b) * (w_t - 2 alpha_t),
    with the planes dv_bitplane_plan derives (gaps wider than 127 split, int8 coefficients)."""
    import ctypes
    from navsim_amd import _native as N
    lib = N.load()
    rng = np.random.default_rng(7)
    level_sets = [[0, 63, 127, 191, 255], [1, 128, 255], [0, 255], [17], [0, 1, 2, 3], list(range(0, 256, 17)),
                  sorted(rng.choice(256, 12, replace=False).tolist())]
    for levels in level_sets:
        presence = (ctypes.c_uint32 * 8)()
        for v in levels:
            presence[v >> 5] |= 1 << (v & 31)
        lo = (ctypes.c_uint8 * 64)()
        w = (ctypes.c_uint8 * 64)()
        lmin, lmax = ctypes.c_int(), ctypes.c_int()
        n = lib.dv_bitplane_plan(presence, 64, lo, w, ctypes.byref(lmin), ctypes.byref(lmax))
        assert n >= 0 and (lmin.value, lmax.value) == (levels[0], levels[-1])
        lo_a, w_a = np.array(lo[:n], dtype=int), np.array(w[:n], dtype=int)
        assert (w_a >= 1).all() and (w_a <= 127).all()
        assert n == sum(-(-(b - a) // 127) for a, b in zip(levels, levels[1:]))
        a = np.arange(256)[:, None]                                   # every patch byte
        b = np.array(levels)[None, :]                                 # every library byte
        alpha = np.clip(a[:, :, None] - lo_a[None, None, :], 0, w_a[None, None, :])           # [a, 1, t]
        bits = (b[:, :, None] >= (lo_a + w_a)[None, None, :]).astype(int)                       # [1, b, t]
        coef = w_a[None, None, :] - 2 * alpha
        assert n == 0 or (coef.min() >= -127 and coef.max() <= 127)
        const = np.maximum(levels[0] - a, 0) + np.maximum(a - levels[-1], 0) + alpha.sum(axis=2)
        got = const + (bits * coef).sum(axis=2)
        assert np.array_equal(got, np.abs(a - b)), levels
    # too many planes for the cap: refused, the byte-plane kernels keep such a library
    presence = (ctypes.c_uint32 * 8)(*([0xffffffff] * 8))
    assert lib.dv_bitplane_plan(presence, 16, lo, w, ctypes.byref(lmin), ctypes.byref(lmax)) == -1


def test_fp4_form_identity_and_plan():
    """The fp4 form of the matrix-core kernel (csrc/dejavu_kernels.h, fp4_segment), on the CPU: for a patch byte ON a level
    (or outside the library's range) every coefficient w_t - 2 alpha_t is +w_t or -w_t, so with the library bit of a plane on
    bit b of a nibble read as the E2M1 value 0.5 / 1 / 2 (bit 3: shifted to 1) and the coefficient as E2M1 +-1,
        sum_t bit_t (w_t - 2 alpha_t) = sum_b wacc[b] * scale_b * (fp32 sum of E2M1 products on bit b)
    exactly -- with the widths dv_fp4_plan derives (the first plane of a gap the int8 form split stands for the whole gap)."""
    import ctypes
    from navsim_amd import _native as N
    lib = N.load()
    e2m1 = {0x0: 0.0, 0x1: 0.5, 0x2: 1.0, 0x4: 2.0, 0xA: -1.0}
    cases = [([0, 63, 127, 191, 255], True), ([0, 85, 170, 255], True), ([0, 255], True), ([1, 128, 255], True), ([0, 60, 130, 255], False),
             ([0, 127, 255], False), ([3, 200], True), ([5], True), ([0, 10, 20, 30, 40, 50, 60, 70, 80], True)]
    for levels, ok in cases:
        presence = (ctypes.c_uint32 * 8)()
        for v in levels:
            presence[v >> 5] |= 1 << (v & 31)
        lo = (ctypes.c_uint8 * 64)()
        w = (ctypes.c_uint8 * 64)()
        lmin, lmax = ctypes.c_int(), ctypes.c_int()
        T = lib.dv_bitplane_plan(presence, 64, lo, w, ctypes.byref(lmin), ctypes.byref(lmax))
        wfull = (ctypes.c_uint8 * 64)()
        wacc = (ctypes.c_int * 4)()
        assert lib.dv_fp4_plan(presence, T, lo, w, wfull, wacc) == (1 if ok else 0), levels
        firsts = [t for t in range(T) if wfull[t]]
        assert [lo[t] for t in firsts] == levels[:-1] and [wfull[t] for t in firsts] == [b - a for a, b in zip(levels, levels[1:])]
        if not ok or T == 0:
            continue
        scale = {0: 2.0, 1: 1.0, 2: 0.5, 3: 1.0}                    # bit 0 stands for 0.5, bit 1 for 1, bit 2 for 2, bit 3 (shifted) for 1
        on_level = sorted(set(levels) | {0, 255, max(levels[0] - 1, 0), min(levels[-1] + 1, 255)} - set(range(levels[0] + 1, levels[-1])) | set(levels))
        for a in on_level:                                          # patch byte: a level, or outside [lmin, lmax]
            if levels[0] < a < levels[-1] and a not in levels:
                continue
            for b in levels:                                        # library byte
                n_px = 5                                            # a few pixels so that planes land on every bit position
                acc = [0.0, 0.0, 0.0, 0.0]
                want = 0
                for n in range(n_px * T):
                    t, bit = n % T, n % 4
                    lib_bit = 1 if b >= lo[t] + w[t] else 0
                    alpha = min(max(a - lo[t], 0), w[t])
                    want += lib_bit * (w[t] - 2 * alpha)            # the int8 form's term
                    if not wfull[t]:
                        continue                                    # a copy of a split gap: coefficient 0 in the fp4 image
                    assert a <= lo[t] or a >= lo[t] + wfull[t]
                    sign = e2m1[0x2] if a <= lo[t] else e2m1[0xA]
                    value = {0: e2m1[0x1], 1: e2m1[0x2], 2: e2m1[0x4], 3: e2m1[0x2]}[bit] if lib_bit else 0.0
                    acc[bit] += sign * value
                got = sum(int(wacc[bit]) * int(scale[bit] * acc[bit]) for bit in range(4))
                assert got == want, (levels, a, b)
    # the 3-bit level code of a five-level plane (k_bitpack_code) and its decode in the kernel
    code = {0: 0b000, 1: 0b001, 2: 0b010, 3: 0b110, 4: 0b111}
    for level, cbits in code.items():
        b0, b1, b2 = cbits & 1, (cbits >> 1) & 1, (cbits >> 2) & 1
        assert [b0 | b1, b1, b2, b0 & b2] == [int(level >= t) for t in (1, 2, 3, 4)]


def test_ssd_u8_expansion_is_exact():
    """What k_ssd_u8_mfma rests on (csrc/dejavu_kernels.h, "ssd_u8 metric"): with a' = a - 128 stored as the byte a ^ 0x80 read as
    an int8, sum (a-b)^2 = sum a'^2 + sum b'^2 - 2 sum a'b' in integers, equal to the oracle's ssds (navsim/util.pyx:171-184) on the
    same uint8 data; and the cross term of the largest allowed patch (131 071 pixels of extreme bytes) still fits an int32."""
    rng = np.random.default_rng(12)
    for shape in ((1, 1), (7, 5), (33, 31), (64, 64)):
        a = rng.integers(0, 256, shape, dtype=np.uint8)
        b = rng.integers(0, 256, shape, dtype=np.uint8)
        if a.size > 2:
            a.flat[0], b.flat[0], a.flat[1], b.flat[1] = 0, 255, 255, 0
        a8 = (a ^ 0x80).view(np.int8).astype(np.int64)
        b8 = (b ^ 0x80).view(np.int8).astype(np.int64)
        assert np.array_equal(a8, a.astype(np.int64) - 128)
        got = (a8 * a8).sum() + (b8 * b8).sum() - 2 * (a8 * b8).sum()
        assert got == oracle.ssds(a.astype(np.float64), b.astype(np.float64)) == ((a.astype(np.int64) - b.astype(np.int64)) ** 2).sum()
    assert 131071 * 128 * 128 <= 2 ** 31 - 1 < 131072 * 128 * 128      # |sum a'b'| <= P * 2^14 (both bytes 0 everywhere): P <= 131 071


def test_experiment_helpers_match_the_reference(manifest, golden):
    """scripts/run_experiment.py's helpers, pinned by the reference's own outputs (tests/golden/make_golden.py imports
    the script; t7_experiment.npz): training-path generator and path chopping bit for bit, the result row of
    run_experiment, and the CSV header / row text of the farm's task files."""
    from navsim_amd import experiment
    z = golden("t7_experiment.npz")
    t7 = manifest["t7_experiment"]
    assert (experiment.FRAME_FACTOR, experiment.N_CONSECUTIVE_SCENES) == (t7["frame_factor"], t7["n_consecutive_scenes"])
    for c in t7["paths"]:
        got = experiment.sin_training_path(c["curve"], c["start_x"], c["l"], arclen=c["arclen"])
        assert got.tobytes() == z[c["key"]].tobytes() and got.shape == z[c["key"]].shape, c["key"]
    for c in t7["chops"]:
        got = experiment.chop_path_to_len(z[c["path"]], c["length"])
        assert got.tobytes() == z[c["key"]].tobytes() and got.shape == z[c["key"]].shape, c["key"]
    land = synth.synth_landscape(t7["landscape"]["seed"], t7["landscape"]["size"], t7["landscape"]["grain"])
    assert sha(land) == t7["landscape"]["sha"]
    size = t7["landscape"]["size"]
    tp = synth.sin_training_path(0.5, 0.2 * size, 0.6 * size, arclen=1.0)[:t7["n_views"]]
    for row in t7["rows"]:
        nsf = navsim_amd.NavBySceneFamiliarity(land, (16, 8), 1.0, n_test_angles=10, sensor_pixel_dimensions=[2, 4],
                                               n_sensor_levels=4, mask_middle_n=1, saccade_degrees=90.0,
                                               max_distance_to_training_path=450,
                                               familiarity_model=oracle.sads_familiarity(row["chem_weight"]))
        nsf.train_from_path(tp)
        d = tp[2] - tp[1]
        nsf.angle = float(np.arctan2(d[1], d[0]) % (2 * np.pi)) + np.deg2rad(7.0)
        nsf.position = tp[1] + np.array([1.5, -1.0])
        res = experiment.run_experiment(nsf, frames=row["frames"])
        assert {k: (int(v) if isinstance(v, (int, np.integer)) else float(v)) for k, v in res.items()} == row["result"], row["name"]
        assert experiment.csv_header(row["trial"]) == row["header"]
        assert experiment.csv_row(row["trial"], res) == row["line"], row["name"]


def test_sanitizer_build_of_the_host_side_runs_clean():
    """AddressSanitizer + UndefinedBehaviorSanitizer on the C code that needs no GPU (tools/sanitize): the host-arithmetic
    entry points of the C ABI (csrc/dejavu_host.inl) and the oracle's C restatement, on exact-size buffers and ragged
    shapes.  (GPU sanitizers are not available on the pool.)"""
    import shutil
    import subprocess
    if shutil.which("g++") is None or shutil.which("make") is None:
        pytest.skip("no host C++ toolchain")
    d = os.path.join(REPO, "tools", "sanitize")
    r = subprocess.run(["make", "-C", d, "check"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "asan_driver: ok" in r.stdout and "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr


def test_batch_results_arrays_are_the_records():
    """engine.BatchResults: the ensemble step's records as arrays and as the per-agent dictionaries -- the same memory."""
    from navsim_amd import _native as N
    from navsim_amd.engine import BatchResults
    n, A = 7, 5
    raw = (N.StepResult * n)()
    for i in range(n):
        raw[i].best_heading, raw[i].flags, raw[i].best_view, raw[i].best_fam = i % A, 16 * (i == 3), 1000 + i, 0.5 * i
        raw[i].n_headings, raw[i].n_candidates = A, i
        for a in range(A):
            raw[i].angle_fam[a], raw[i].angle_view[a] = i + 0.25 * a, 10 * i + a
    res = BatchResults(raw, n, A)
    assert len(res) == n and res.angle_familiarity.shape == (n, A) and len(list(res)) == n and len(res[1:4]) == 3
    assert res.best_idex.tolist() == [i % A for i in range(n)] and res.flags.tolist() == [0, 0, 0, 16, 0, 0, 0]
    for i in range(n):
        d = res[i]
        assert (d["best_idex"], d["best_view"], d["step_familiarity"], d["flags"], d["n_candidates"]) == \
               (res.best_idex[i], res.best_view[i], res.step_familiarity[i], res.flags[i], res.n_candidates[i])
        assert np.array_equal(d["angle_familiarity"], res.angle_familiarity[i]) and np.array_equal(d["angle_view"], res.angle_view[i])
    assert res[-1]["best_view"] == 1000 + n - 1
    with pytest.raises(IndexError):
        res[n]


def test_deferring_the_error_metrics_is_bounded_by_the_triangle_inequality():
    """agent._can_defer_error: with a finite max_distance_to_training_path the distance of the position a step ends at may be collected
    a step late only while (last known distance) + (way since it was measured) + step_size cannot exceed the limit."""
    land = synth.synth_landscape(1, 120, 4)
    nsf = navsim_amd.NavBySceneFamiliarity(land, (8, 8), 1.0, n_test_angles=4, familiarity_model=oracle.sads_familiarity(0.0),
                                           use_gpu_sensor=False, max_distance_to_training_path=10.0)
    nsf.train_from_path(np.stack([np.linspace(30, 90, 40), np.full(40, 60.0)], axis=1))
    assert not nsf._metrics_on_device and not nsf._can_defer_error((50.0, 60.0))      # host metrics: nothing to defer
    nsf._metrics_on_device = True
    assert not nsf._can_defer_error((50.0, 60.0))                                       # no distance known yet
    nsf._last_nearest = (3.0, (50.0, 60.0))
    assert nsf._can_defer_error((55.0, 60.0))                                           # 3 + 5 + 1 <= 10
    assert nsf._can_defer_error((53.0, 64.0))                                           # 3 + 5 + 1 (a 3-4-5 way)
    assert not nsf._can_defer_error((56.5, 60.0))                                       # 3 + 6.5 + 1 > 10
    assert not nsf._can_defer_error((56.0, 60.0))                                       # exactly 10: the margin decides against
    nsf._last_nearest = (3.0, None)
    assert not nsf._can_defer_error((50.0, 60.0))
    nsf.max_distance_to_training_path = np.inf
    assert nsf._can_defer_error((1e9, 1e9))
    nsf._metrics_on_device = False
