"""One process, several devices (dv_group_*, navsim_amd.FamiliarityGroup; SURVEY 8-b2's dv_create(device_ids, n)): the library cut
into blocks over n member contexts -- here all on GPU 0, which is what a one-GPU box offers; the members are independent contexts
either way -- must give the reference's UNSHARDED decision: the golden step vectors (tie stress included), the per-view minimum in
library order, the plug-in's func(scene, fambuf), device-sensed steps, and an agent walking with the plug-in made from a device list.
Everything calls libdejavu_hip.so through ctypes; the oracle is the checker only.
"""
import numpy as np
import pytest

import navsim_amd
from navsim_amd import synth
from oracle import oracle
from tests.helpers import step_case_inputs

pytestmark = pytest.mark.gpu

RTOL = 1e-9


@pytest.mark.parametrize("n", [1, 2, 3, 5])
def test_group_steps_equal_the_golden_vectors(n, manifest, golden):
    z = golden("t2_step.npz")
    group = navsim_amd.FamiliarityGroup([0] * n)
    try:
        assert len(group) == n
        resolved = 0
        for case in manifest["t2_step"]:
            lib, patches = step_case_inputs(case)
            if len(lib) < n:
                continue
            name = case["name"]
            group.set_library(lib, case["chem_weight"])
            lo_hi = [group.bounds(r) for r in range(n)]
            assert lo_hi[0][0] == 0 and lo_hi[-1][1] == len(lib) and all(a[1] == b[0] for a, b in zip(lo_hi, lo_hi[1:]))
            r = group.step(patches, want_scene=True)
            assert r["best_idex"] == case["best_idex"], (name, n, r["n_candidates"], r["flags"])
            assert r["best_view"] == case["best_view"], (name, n)
            np.testing.assert_allclose(r["angle_familiarity"], z[name + "_angle"], rtol=RTOL, err_msg=name)
            np.testing.assert_allclose(r["scene_familiarity"], z[name + "_scene"], rtol=RTOL, err_msg=name)
            np.testing.assert_allclose(r["step_familiarity"], case["step_familiarity"], rtol=RTOL)
            if r["resolved"]:
                resolved += 1
                assert r["step_familiarity"] == case["step_familiarity"], name      # exact values decided: the reference's double
            # every heading's first view holds that heading's maximum
            fam = np.array([oracle.sads_hsv(lib[int(v)][None], patches[a], case["chem_weight"])[0] for a, v in enumerate(r["angle_view"])])
            np.testing.assert_allclose(fam, z[name + "_angle"], rtol=RTOL)
            # the plug-in's call: one heading, all views, in library order
            buf = np.full(len(lib), np.nan)
            group.score(patches[0], buf)
            one = np.empty(len(lib))
            oracle.sads_familiarity(case["chem_weight"])(lib)(patches[0], one)
            np.testing.assert_allclose(buf, one, rtol=RTOL)
        assert n == 1 or resolved > 0                                              # the tie fixtures did go through the members' resolvers
    finally:
        group.close()


def test_group_ties_across_members_take_the_first_view():
    """Exact duplicates of the best view in different members' blocks, seen by two headings: first heading, lowest view."""
    lib = synth.synth_views(21, 300, 8, 8)
    pat = synth.synth_patches(21, 6, 8, 8)
    lib[10] = lib[290] = lib[150]
    pat[1] = lib[150]
    pat[4] = lib[150]
    want = oracle.step(lib, pat, 0.0)
    assert (want["best_idex"], want["best_view"]) == (1, 10)
    for n in (2, 3, 4):
        group = navsim_amd.FamiliarityGroup([0] * n)
        try:
            group.set_library(lib, 0.0)
            r = group.step(pat, want_scene=False)
            assert (r["best_idex"], r["best_view"], r["step_familiarity"]) == (1, 10, 64.0), n
            assert r["scene_familiarity"] is None
        finally:
            group.close()


def test_group_sense_step_equals_one_engine():
    """dv_group_sense_step: every member senses from its own copy of the landscape; the decision is the single context's."""
    land = synth.synth_landscape(9, 300, 4)
    path = synth.sin_training_path(0.5, 60, 200, arclen=1.0)[:150]
    nsf = navsim_amd.NavBySceneFamiliarity(land, (12, 12), 1.0, n_test_angles=10, n_sensor_levels=5,
                                           familiarity_model=navsim_amd.sads_familiarity(0.25))
    nsf.train_from_path(path)
    group = navsim_amd.FamiliarityGroup([0, 0, 0])
    try:
        group.set_landscape(land)
        group.configure_sensor(nsf.sensor_dimensions, nsf.sensor_pixel_dimensions, nsf._level_tables(), nsf.mask_middle_n)
        group.set_library(nsf.familiar_scenes, 0.25)
        rng = np.random.default_rng(4)
        for k in range(12):
            i = int(rng.integers(3, len(path) - 3))
            x, y = path[i] + rng.uniform(-0.6, 0.6, 2)
            angles = (rng.uniform(0, 2 * np.pi) + nsf.angle_offsets) % (2 * np.pi)
            one = nsf._engine.sense_step(x, y, angles, want_scene=True)
            got = group.sense_step(x, y, angles, want_scene=True)
            assert (got["best_idex"], got["best_view"]) == (one["best_idex"], one["best_view"]), k
            np.testing.assert_allclose(got["angle_familiarity"], one["angle_familiarity"], rtol=RTOL)
            np.testing.assert_allclose(got["scene_familiarity"], one["scene_familiarity"], rtol=RTOL)
        with pytest.raises(IndexError):                                            # a footprint past the landscape: the reference's IndexError
            group.sense_step(299.4, 299.4, angles)              # (negative indices wrap as in NumPy; indices past the end raise)
    finally:
        group.close()
        nsf.clear_training()


def test_agent_with_the_plugin_made_from_a_device_list():
    """sads_familiarity(cw, devices=[...]) as the agent's familiarity_model: the trajectory of the oracle-scored agent."""
    land = synth.synth_landscape(8, 400, 4)
    path = synth.sin_training_path(0.5, 60, 260, arclen=1.0)[:200]
    kw = dict(n_test_angles=9, n_sensor_levels=5)
    dev = navsim_amd.NavBySceneFamiliarity(land, (16, 16), 1.0, familiarity_model=navsim_amd.sads_familiarity(0.25, devices=[0, 0, 0]), **kw)
    ref = navsim_amd.NavBySceneFamiliarity(land, (16, 16), 1.0, familiarity_model=oracle.sads_familiarity(0.25), use_gpu_sensor=False, **kw)
    for nsf in (dev, ref):
        nsf.train_from_path(path)
        nsf.position, nsf.angle = (path[2][0] + 0.6, path[2][1] - 0.3), 0.8
    assert isinstance(dev._familiarity_func.engine, navsim_amd.FamiliarityGroup)
    assert dev._familiarity_func.engine.sensor_attached          # the members sense the patches themselves: only the pose goes up
    for t in range(60):
        dev.step_forward()
        ref.step_forward()
        assert dev.last_best_idex == ref.last_best_idex, t
        assert dev.position == ref.position and dev.angle == ref.angle, t
        np.testing.assert_allclose(dev.angle_familiarity, ref.angle_familiarity, rtol=RTOL)
        np.testing.assert_allclose(dev.scene_familiarity, ref.scene_familiarity, rtol=RTOL)
    assert dev.navigation_error == ref.navigation_error
    dev._familiarity_func.engine.close()


def test_group_errors():
    with pytest.raises(ValueError):
        navsim_amd.FamiliarityGroup([])
    with pytest.raises(navsim_amd.EngineError):
        navsim_amd.FamiliarityGroup([0, 99])                                       # no such device: nothing is left behind
    group = navsim_amd.FamiliarityGroup([0, 0])
    try:
        with pytest.raises((ValueError, navsim_amd.EngineError)):
            group.step(np.zeros((3, 8, 8, 3), np.uint8))                           # no library
        with pytest.raises(ValueError):
            group.set_library(synth.synth_views(1, 1, 8, 8), 0.0)                  # fewer views than members
        group.set_library(synth.synth_views(1, 40, 8, 8), 0.0)
        with pytest.raises(ValueError):
            group.step(np.zeros((3, 8, 9, 3), np.uint8))
        with pytest.raises(ValueError):
            group.score(np.zeros((8, 8, 3), np.uint8), np.zeros(40, np.float32))
    finally:
        group.close()
