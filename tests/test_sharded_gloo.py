"""Library sharding across ranks (navsim_amd/sharded.py) with world_size 2 and 3 over gloo, on CPU.

Each rank scores its own block of views through a test double of the engine (tests/fake_engine.py,
oracle-backed); the exchange, the cross-rank tie protocol and the merge are the product code.  The
merged decision must equal the reference's unsharded one (oracle.step) bit for bit.
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from navsim_amd import sharded, synth
from oracle import oracle
from tests.fake_engine import OracleBackedEngine


def cases():
    out = []
    lib = synth.synth_views(21, 300, 8, 8)
    pat = synth.synth_patches(21, 6, 8, 8)
    out.append(("plain", lib, pat, 0.0))
    out.append(("plain_cw", lib, pat, 0.5))
    # exact duplicates of the best view on BOTH sides of the shard boundary, seen by two headings
    lib2 = lib.copy()
    lib2[10] = lib2[290] = lib2[150]
    pat2 = pat.copy()
    pat2[1] = lib2[150]
    pat2[4] = lib2[150]
    out.append(("dups_across_ranks", lib2, pat2, 0.0))
    # near-ties: equal integer SAD, different pixel order -> ulp-different doubles decide
    small = synth.synth_views(5, 4000, 4, 4)
    out.append(("ties_4x4", small, synth.synth_patches(5, 16, 4, 4), 0.0))
    out.append(("ties_4x4_cw", small, synth.synth_patches(5, 16, 4, 4), 0.3))
    # everything identical: every pair is a candidate on every rank
    same = np.repeat(lib[:1], 50, axis=0)
    out.append(("all_same", same, np.repeat(lib[:1], 5, axis=0), 0.25))
    return out


def worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    gather = sharded.torch_gather(device=None)
    reduce_max = sharded.torch_reduce_max(device=None)
    results = []
    for name, lib, pat, cw in cases():
        sh = sharded.ShardedFamiliarity(OracleBackedEngine(), gather, rank, world)
        sh.set_library(lib, cw)
        lo, hi = sh.bounds
        assert (lo, hi) == sharded.shard_bounds(len(lib), world, rank)
        r = sh.step(pat)
        results.append((name, r["best_idex"], r["best_view"], r["step_familiarity"],
                        r["angle_familiarity"].tolist(), sh.exchanges))
        # the same step with the fast exchange first: one all-reduce(max) of packed keys, full records only on near-ties
        shk = sharded.ShardedFamiliarity(OracleBackedEngine(), gather, rank, world, reduce_max=reduce_max)
        shk.set_library(lib, cw)
        rk = shk.step(pat)
        results.append((name + "+keys", rk["best_idex"], rk["best_view"], rk["step_familiarity"],
                        rk["angle_familiarity"].tolist(), (shk.exchanges, shk.key_decisions)))
        # the per-view minimum stays sharded and is gathered when read: every rank ends with the unsharded array
        with pytest.raises(RuntimeError):
            shk.gather_scene_familiarity()
        rs = shk.step(pat, want_scene=True)
        assert rs["scene_familiarity"] is None and len(rs["scene_familiarity_local"]) == hi - lo
        results.append((name + "+scene", rs["best_idex"], rs["best_view"], rs["step_familiarity"],
                        shk.gather_scene_familiarity().tolist(), 0))
    q.put((rank, results))
    dist.barrier()
    dist.destroy_process_group()


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_matches_unsharded_reference(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    expected = {name: oracle.step(lib, pat, cw) for name, lib, pat, cw in cases()}
    for name in list(expected):
        expected[name + "+keys"] = expected[name]
        expected[name + "+scene"] = dict(expected[name], angle_familiarity=expected[name]["scene_familiarity"])
    exchanged = {}
    for rank in range(world):
        for name, best, view, fam, angle, exchanges in got[rank]:
            want = expected[name]
            assert best == want["best_idex"], (name, rank)
            assert view == want["best_view"], (name, rank)
            np.testing.assert_allclose(fam, want["step_familiarity"], rtol=1e-12)
            np.testing.assert_allclose(angle, want["angle_familiarity"], rtol=1e-12)
            exchanged[name] = exchanges
    # one exchange when the integer scores decide, a second one only for cross-rank ties
    assert exchanged["plain"] == 1 and exchanged["plain_cw"] == 1
    assert exchanged["dups_across_ranks"] >= 1
    # the key exchange alone decides the plain cases (1 collective of A + 4*world words); ties go on to the records
    assert exchanged["plain+keys"] == (1, 1) and exchanged["plain_cw+keys"] == (1, 1)
    for name in ("dups_across_ranks", "all_same"):               # (ties_4x4's near-ties need not involve the maximum)
        n_ex, n_key = exchanged[name + "+keys"]
        assert n_key == 0 and n_ex >= 2, name


def test_key_merge_rules_native_equals_python():
    """dv_merge_keys (host C in the library) against sharded.merge_keys on random reduced keys, incl. the signed-order
    transport and the sense-error word."""
    rng = np.random.default_rng(3)
    delta = 1e-9
    n_decided = 0
    for it in range(300):
        world, A = int(rng.integers(1, 9)), int(rng.integers(1, 65))
        recs = []
        for r in range(world):
            ang = rng.choice([10.0, 10.0 + 1e-12, 9.5, 3.25, 7.0, -1.5], A) + (0.0 if it % 3 else rng.uniform(0, 1e-3, A))
            rec = np.empty(3 + 4 * A)
            rec[0] = ang.max()
            rec[1] = float(rng.integers(1, 4) if it % 2 else 1)
            rec[2] = float(rng.integers(0, 3))
            rec[3:3 + A] = ang
            rec[3 + A:3 + 2 * A] = rng.integers(0, 1 << 33, A)
            rec[3 + 2 * A:] = 0.0
            recs.append(rec)
        keys = np.zeros(A + 4 * world, dtype=np.uint64)
        for r in range(world):
            keys = np.maximum(keys, sharded.pack_keys(recs[r], r, world))
        want = sharded.merge_keys(keys, world, A, delta)
        for signed in (False, True):
            wire = keys ^ np.uint64(1 << 63) if signed else keys
            got = sharded.merge_keys_native(wire, world, A, delta, signed_order=signed)
            assert (got is None) == (want is None), it
            if want is not None:
                n_decided += 1
                assert (got["best_idex"], got["best_view"], got["step_familiarity"]) == (
                    want["best_idex"], want["best_view"], want["step_familiarity"]), it
                assert np.array_equal(got["angle_familiarity"], want["angle_familiarity"])
                # and it is what the full records would have decided
                full = sharded.merge_records(np.stack(recs), delta, A)
                assert (full["best_idex"], full["best_view"], full["step_familiarity"]) == (
                    want["best_idex"], want["best_view"], want["step_familiarity"]), it
    assert n_decided > 100
    rec = recs[0].copy()
    rec[2] += 4.0                                           # sensed past the end of the landscape
    keys = sharded.pack_keys(rec, 0, 1)
    with pytest.raises(IndexError):
        sharded.merge_keys(keys, 1, A, delta)
    with pytest.raises(IndexError):
        sharded.merge_keys_native(keys, 1, A, delta)


def ensemble_case():
    lib = synth.synth_views(33, 200, 8, 8)
    pats = synth.synth_patches(33, 5 * 4, 8, 8).reshape(5, 4, 8, 8, 3).copy()
    pats[3, 2] = lib[77]                       # agent 3 stands on a stored view
    pats[0, 1] = pats[0, 3] = lib[5]           # agent 0: two headings tie exactly -> first one wins
    return lib, pats, 0.25


def ensemble_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lib, pats, cw = ensemble_case()
    ens = sharded.ShardedEnsemble(OracleBackedEngine(), rank, world)
    ens.set_library(lib, cw)
    res = ens.step(pats)
    lo, hi = ens.agent_bounds(len(pats))
    assert len(res) == hi - lo
    table = ens.decisions(res, len(pats), sharded.torch_gather(device=None))
    q.put((rank, (lo, hi), table.tolist()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_agent_sharded_ensemble_matches_reference(world):
    """BASELINE configs[4]: agents partitioned over ranks (5 agents on 2 or 3 ranks: uneven blocks), library replicated."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=ensemble_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    lib, pats, cw = ensemble_case()
    want = [oracle.step(lib, p, cw) for p in pats]
    spans = sorted(g[1] for g in got)
    assert spans[0][0] == 0 and spans[-1][1] == len(pats) and all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
    for rank, _, table in got:
        for g, w in enumerate(want):
            assert int(table[g][0]) == w["best_idex"] and int(table[g][1]) == w["best_view"], (rank, g)
            np.testing.assert_allclose(table[g][2], w["step_familiarity"], rtol=1e-12)
    assert want[3]["best_idex"] == 2 and want[3]["best_view"] == 77 and want[0]["best_idex"] == 1


def trajectory_worker(rank, world, port, q):
    """The agent itself on a sharded library: every rank runs the same NavBySceneFamiliarity, decisions come through
    the exchange (multi-GPU result identical to single-GPU, here against the reference's trajectory)."""
    import json
    import navsim_amd
    from tests.test_host_logic import _run_trajectory
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    here = os.path.dirname(os.path.abspath(__file__))
    case = [c for c in json.load(open(os.path.join(here, "golden", "manifest.json")))["t4_trajectory"] if c["name"] == "traj_px"][0]
    land = synth.synth_landscape(case["landscape"]["seed"], case["landscape"]["size"], case["landscape"]["grain"])
    model = sharded.sharded_sads_familiarity(case["chem_weight"], sharded.torch_gather(device=None), rank, world,
                                             engine_factory=OracleBackedEngine)
    nsf, best, pos, ang, fam, status = _run_trajectory(case, land, model, use_gpu_sensor=False, track_scene_familiarity=False)
    q.put((rank, best.tolist(), pos.tobytes(), ang.tobytes(), fam.tolist(), status, nsf._familiarity_func.engine.exchanges))
    dist.barrier()
    dist.destroy_process_group()


def test_agent_trajectory_on_a_sharded_library_matches_the_reference():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=trajectory_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=600) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "t4_trajectory.npz"))
    for rank, best, pos, ang, fam, status, exchanges in got:
        assert status == 0 and len(best) == 250
        assert np.array_equal(np.array(best), z["traj_px_best"]), rank           # heading index: bit-identical
        assert pos == z["traj_px_pos"].tobytes() and ang == z["traj_px_angle"].tobytes()
        np.testing.assert_allclose(fam, z["traj_px_fam"], rtol=1e-12, atol=0)
        assert exchanges >= 250


def test_shard_bounds_cover_the_library():
    for n, w in ((10, 3), (50000, 8), (7, 8), (64, 2)):
        spans = [sharded.shard_bounds(n, w, r) for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
        sizes = [hi - lo for lo, hi in spans]
        assert max(sizes) - min(sizes) <= 1


def test_merge_rules_on_hand_made_records():
    A = 3
    def rec(approx_max, ncand, state, ang, view, ex=None, exv=None):
        r = np.empty(3 + 4 * A)
        r[0:3] = approx_max, ncand, state
        r[3:3 + A] = ang
        r[3 + A:3 + 2 * A] = view
        r[3 + 2 * A:3 + 3 * A] = ex if ex is not None else [-np.inf] * A
        r[3 + 3 * A:3 + 4 * A] = exv if exv is not None else [-1] * A
        return r
    delta = 1e-9
    # single candidate overall: integer scores decide
    recs = np.stack([rec(10.0, 1, 0, [10.0, 7.0, 8.0], [5, 6, 7]), rec(9.0, 1, 0, [9.0, 8.5, 6.0], [105, 106, 107])])
    again, ranks = sharded.needs_resolve(recs, delta)
    assert not again and ranks == [0]
    m = sharded.merge_records(recs, delta, A)
    assert m["best_idex"] == 0 and m["best_view"] == 5 and list(m["angle_familiarity"]) == [10.0, 8.5, 8.0]
    # both ranks hold the maximum: must resolve; after resolving, exact values decide, first heading wins ties
    recs = np.stack([rec(10.0, 1, 0, [7.0, 10.0, 8.0], [5, 6, 7]), rec(10.0, 1, 0, [10.0, 8.5, 6.0], [105, 106, 107])])
    again, ranks = sharded.needs_resolve(recs, delta)
    assert again and ranks == [0, 1]
    recs = np.stack([rec(10.0, 1, 1, [7.0, 10.0, 8.0], [5, 6, 7], [-np.inf, 10.0, -np.inf], [-1, 6, -1]),
                     rec(10.0, 1, 1, [10.0, 8.5, 6.0], [105, 106, 107], [10.0, -np.inf, -np.inf], [105, -1, -1])])
    assert not sharded.needs_resolve(recs, delta)[0]
    m = sharded.merge_records(recs, delta, A)
    assert m["best_idex"] == 0 and m["best_view"] == 105 and m["resolved"]
    with pytest.raises(RuntimeError):
        sharded.merge_records(np.stack([rec(10.0, 2, 0, [10.0, 1, 1], [1, 2, 3]), rec(10.0, 1, 0, [10.0, 1, 1], [4, 5, 6])]),
                              delta, A)


def test_native_merge_matches_python_rules():
    """dv_merge_records (host C in the library, used by DeviceExchange) against the Python statement of the rules."""
    rng = np.random.default_rng(7)
    n_again = n_exact = n_plain = 0
    for trial in range(600):
        world = int(rng.choice([1, 2, 3, 8]))
        A = int(rng.choice([1, 5, 16]))
        delta = 1e-9
        rec = np.zeros((world, 3 + 4 * A + int(rng.integers(0, 3))))       # stride may exceed the record
        levels = np.array([100.0, 100.0 + 5e-10, 100.0 - 4e-10, 99.0, 42.5])
        for r in range(world):
            ang = rng.choice(levels, size=A)
            rec[r, 3:3 + A] = ang
            rec[r, 0] = ang.max()
            rec[r, 1] = rng.choice([1, 1, 1, 2, 5])
            rec[r, 2] = rng.choice([0.0, 1.0, 1.0, 2.0])
            rec[r, 3 + A:3 + 2 * A] = rng.integers(0, 1000, A) + 1000 * r
            ex = np.where(rng.random(A) < 0.5, ang + rng.choice([0.0, 1e-13, -1e-13], size=A), -np.inf)
            ex[int(np.argmax(ang))] = ang.max() + rng.choice([0.0, 1e-13, -1e-13])
            rec[r, 3 + 2 * A:3 + 3 * A] = ex
            rec[r, 3 + 3 * A:3 + 4 * A] = np.where(np.isfinite(ex), rng.integers(0, 1000, A) + 1000 * r, -1)
        again_py, ranks_py = sharded.needs_resolve(rec, delta)
        again_c, ranks_c, out_c = sharded.merge_records_native(rec, delta, A)
        assert again_c == again_py and ranks_c == ranks_py, trial
        if again_py:
            n_again += 1
            assert out_c is None
            continue
        out_py = sharded.merge_records(rec, delta, A)
        n_exact += out_py["resolved"]
        n_plain += not out_py["resolved"]
        assert out_c["best_idex"] == out_py["best_idex"] and out_c["best_view"] == out_py["best_view"], trial
        assert out_c["step_familiarity"] == out_py["step_familiarity"] and out_c["resolved"] == out_py["resolved"], trial
        np.testing.assert_array_equal(out_c["angle_familiarity"], out_py["angle_familiarity"])
    assert n_again > 20 and n_exact > 20 and n_plain > 20, (n_again, n_exact, n_plain)
