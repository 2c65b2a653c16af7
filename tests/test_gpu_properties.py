"""Property tests on the GPU (hypothesis): random small problems, HIP path vs the oracle.

Properties (tier T7): decisions equal the reference's for every input; a permutation of the
library permutes the per-view outputs; sharded == unsharded; integer-score ties never invert the exact order.
"""
import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings, strategies as st

import navsim_amd
from navsim_amd import sharded, synth
from oracle import oracle

from tests.helpers import ENGINE_MODES, engine_mode

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", params=ENGINE_MODES)
def eng(request):
    """Every test runs under each form of the scoring path (tests/helpers.py:ENGINE_MODES): both step endings, the
    bit-plane matrix-core kernel, and the product default."""
    with engine_mode(request.param):
        e = navsim_amd.FamiliarityEngine(device=0)
    e.mode = request.param
    yield e
    e.close()


def build(seed, F, h, w, A, kind):
    lib = synth.random_hsv(seed, (F, h, w, 3))
    pat = synth.random_hsv(seed + 1, (A, h, w, 3))
    if kind == "levels":                         # few values everywhere: many exact integer ties
        lib = synth.synth_views(seed, F, h, w)
        pat = synth.synth_patches(seed, A, h, w)
    elif kind == "two_hues":
        lib[..., 0] = np.where(lib[..., 0] & 1, 9, 200)
        pat[..., 0] = np.where(pat[..., 0] & 1, 9, 200)
        lib[..., 1] >>= (seed & 1)                # half of the cases fit the signed plane (S <= 127)
    elif kind == "few_hues":
        lib[..., 0] %= 4
        pat[..., 0] %= 5
    elif kind == "coarse":                       # 3-level values, heavy ties on larger sensors too
        lib[..., 2] = (lib[..., 2] % 3) * 127
        pat[..., 2] = (pat[..., 2] % 3) * 127
        lib[..., 1] = 0
        pat[..., 1] = 0
    return lib, pat


@settings(max_examples=60, deadline=None, derandomize=True, suppress_health_check=[HealthCheck.function_scoped_fixture, HealthCheck.too_slow])
@given(seed=st.integers(0, 10 ** 6), F=st.integers(1, 300), h=st.integers(1, 12), w=st.integers(1, 12),
       A=st.integers(1, 20), cw=st.sampled_from([0.0, 0.25, 0.5, 1.0, 0.3141592653589793]),
       kind=st.sampled_from(["random", "levels", "two_hues", "few_hues", "coarse"]))
def test_decisions_equal_the_reference(eng, seed, F, h, w, A, cw, kind):
    lib, pat = build(seed, F, h, w, A, kind)
    want = oracle.step(lib, pat, cw)
    eng.set_library(lib, cw)
    got = eng.step(pat, want_scene=True)
    assert got["best_idex"] == want["best_idex"], (got["n_candidates"], got["flags"])
    assert got["best_view"] == want["best_view"]
    np.testing.assert_allclose(got["angle_familiarity"], want["angle_familiarity"], rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(got["scene_familiarity"], want["scene_familiarity"], rtol=1e-9, atol=1e-12)
    if got["flags"] & 3:                          # resolved or exact: the winning score is the reference's double
        assert got["step_familiarity"] == want["step_familiarity"]


@settings(max_examples=15, deadline=None, derandomize=True, suppress_health_check=[HealthCheck.function_scoped_fixture, HealthCheck.too_slow])
@given(seed=st.integers(0, 10 ** 6), F=st.integers(2, 200), cw=st.sampled_from([0.0, 0.4]), parts=st.integers(2, 4))
def test_permutation_and_sharding_invariance(eng, seed, F, cw, parts):
    h, w, A = 6, 7, 5
    lib = synth.synth_views(seed, F, h, w)
    pat = synth.synth_patches(seed, A, h, w)
    eng.set_library(lib, cw)
    base = eng.step(pat, want_scene=True)
    perm = np.random.default_rng(seed).permutation(F)
    eng.set_library(lib[perm], cw)
    shuffled = eng.step(pat, want_scene=True)
    assert np.array_equal(shuffled["scene_familiarity"], base["scene_familiarity"][perm])
    assert np.array_equal(shuffled["angle_familiarity"], base["angle_familiarity"])
    # sharded: contiguous blocks scored separately, merged by the product's record logic
    recs = []
    for r in range(parts):
        lo, hi = sharded.shard_bounds(F, parts, r)
        if hi == lo:
            continue
        eng.set_library(lib[lo:hi], cw, first_view=lo)
        recs.append(sharded.pack_record(eng.step(pat, want_scene=False, force_resolve=True)))
    merged = sharded.merge_records(np.stack(recs), base["delta"], A)
    want = oracle.step(lib, pat, cw)
    assert merged["best_idex"] == want["best_idex"] and merged["best_view"] == want["best_view"]
