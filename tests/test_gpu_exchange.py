"""GPU tests of the multi-rank exchange path at world size 1 (one MI355X): the per-step record on the device, the
hand-over kernel (dv_publish), the RCCL all-gather on the engine's stream and the native merge must reproduce the
plain single-context step, ties included.  The N > 1 protocol itself is covered on CPU over gloo
(tests/test_sharded_gloo.py)."""
import hashlib
import os
import socket

import numpy as np
import pytest

import navsim_amd
from navsim_amd import sharded, synth
from oracle import oracle

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    e = navsim_amd.FamiliarityEngine(0)                  # the product default (nothing forced)
    yield e
    e.close()


def _library_with_ties():
    lib = synth.synth_views(77, 900, 12, 12)
    pats = synth.synth_patches(77, 7, 12, 12)
    lib[500] = lib[40]                      # duplicate views: equal scores, first one must win
    pats[3] = lib[40]
    pats[5] = lib[40]                       # and two headings tie exactly: first heading wins
    return lib, pats


def test_publish_hands_the_step_record_to_the_host(eng):
    lib, pats = _library_with_ties()
    eng.set_library(lib, 0.3)
    eng.upload_patches(pats)
    eng.step_enqueue(want_scene=False)
    res = eng.step_wait(want_scene=False)
    ptr, n = eng.step_record()
    assert n == 3 + 4 * len(pats)
    for _ in range(3):                      # the sequence word advances with every publish
        eng.publish(ptr, n)
        got = eng.publish_wait(np.empty(n))
    np.testing.assert_array_equal(got, sharded.pack_record(res))


def test_device_exchange_single_rank_matches_plain_step(eng):
    torch = pytest.importorskip("torch")
    import torch.distributed as dist
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        for cw in (0.0, 0.3):
            lib, pats = _library_with_ties()
            want = oracle.step(lib, pats, cw)
            eng.set_library(lib, cw)
            eng.upload_patches(pats)
            for direct in ("1", "0"):       # RCCL on the engine's stream, then torch.distributed's collective
                os.environ["DEJAVU_DIRECT_RCCL"] = direct
                ex = sharded.DeviceExchange(eng, 0, 1, torch.device("cuda", 0))
                assert (ex.direct is not None) == (direct == "1")
                for _ in range(3):
                    out = ex.step()
                assert out["best_idex"] == want["best_idex"] == 3
                assert out["best_view"] == want["best_view"] == 40
                np.testing.assert_allclose(out["step_familiarity"], want["step_familiarity"], rtol=1e-12)
                np.testing.assert_allclose(out["angle_familiarity"], want["angle_familiarity"], rtol=1e-12)
                assert ex.exchanges >= 3
                ex.close()
        # the fast exchange: without near-ties ONE all-reduce(max) of packed keys decides the step (k_make_keys,
        # ncclAllReduce(uint64, max) on the engine's stream or torch's int64 maximum with the sign bit flipped),
        # with the same decision as the full records (DEJAVU_KEY_EXCHANGE=0)
        lib = synth.synth_views(78, 1500, 12, 12)
        pats = synth.synth_patches(78, 7, 12, 12)
        pats[2] = synth.near_match_patch(lib[700], 3)
        want = oracle.step(lib, pats, 0.3)
        eng.set_library(lib, 0.3)
        eng.upload_patches(pats)
        for direct in ("1", "0"):
            os.environ["DEJAVU_DIRECT_RCCL"] = direct
            seen = {}
            for keys in ("1", "0"):
                os.environ["DEJAVU_KEY_EXCHANGE"] = keys
                ex = sharded.DeviceExchange(eng, 0, 1, torch.device("cuda", 0))
                for _ in range(4):
                    out = ex.step()
                seen[keys] = (out["best_idex"], out["best_view"], float(out["step_familiarity"]),
                              np.asarray(out["angle_familiarity"]).tolist())
                assert (ex.key_decisions, ex.exchanges) == ((4, 4) if keys == "1" else (0, 4)), (direct, keys)
                ex.close()
            assert seen["1"] == seen["0"]
            assert seen["1"][:2] == (want["best_idex"], want["best_view"]) == (2, 700)
    finally:
        os.environ.pop("DEJAVU_DIRECT_RCCL", None)
        os.environ.pop("DEJAVU_KEY_EXCHANGE", None)
        eng.set_stream(None)
        dist.destroy_process_group()


def test_agent_sharded_ensemble_on_the_engine(eng):
    lib = synth.synth_views(5, 700, 16, 16)
    pats = synth.synth_patches(5, 6 * 8, 16, 16).reshape(6, 8, 16, 16, 3).copy()
    pats[4, 6] = lib[123]
    ens = sharded.ShardedEnsemble(eng, 0, 1)
    ens.set_library(lib, 0.25)
    res = ens.step(pats)
    assert len(res) == 6
    for g, r in enumerate(res):
        want = oracle.step(lib, pats[g], 0.25)
        assert (r["best_idex"], r["best_view"]) == (want["best_idex"], want["best_view"])
    assert (res[4]["best_idex"], res[4]["best_view"]) == (6, 123)


def test_sharded_agent_trajectory_on_the_engine():
    """navsim_amd.NavBySceneFamiliarity over sharded.sharded_sads_familiarity with the real engine (one shard, gather =
    identity): the reference's golden 250-step trajectory, bit for bit."""
    import json
    from tests.test_host_logic import _run_trajectory
    here = os.path.dirname(os.path.abspath(__file__))
    case = [c for c in json.load(open(os.path.join(here, "golden", "manifest.json")))["t4_trajectory"] if c["name"] == "traj_px"][0]
    z = np.load(os.path.join(here, "golden", "t4_trajectory.npz"))
    land = synth.synth_landscape(case["landscape"]["seed"], case["landscape"]["size"], case["landscape"]["grain"])
    model = sharded.sharded_sads_familiarity(case["chem_weight"], lambda rec: rec.reshape(1, -1), 0, 1)
    nsf, best, pos, ang, fam, status = _run_trajectory(case, land, model, use_gpu_sensor=False, track_scene_familiarity=False)
    assert status == 0 and np.array_equal(best, z["traj_px_best"])
    assert pos.tobytes() == z["traj_px_pos"].tobytes() and ang.tobytes() == z["traj_px_angle"].tobytes()
    np.testing.assert_allclose(fam, z["traj_px_fam"], rtol=1e-12, atol=0)
    nsf._familiarity_func.engine.engine.close()


def test_device_sharded_agent_with_gpu_sensor_single_rank():
    """The full multi-GPU agent path at world size 1: landscape, sensor model and library shard on the GPU, every step
    = sense + score + RCCL exchange + merge.  Must reproduce the reference's golden trajectory bit for bit."""
    torch = pytest.importorskip("torch")
    import json
    import torch.distributed as dist
    from tests.test_host_logic import _run_trajectory
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        here = os.path.dirname(os.path.abspath(__file__))
        z = np.load(os.path.join(here, "golden", "t4_trajectory.npz"))
        for case in json.load(open(os.path.join(here, "golden", "manifest.json")))["t4_trajectory"]:
            if case["name"] not in ("traj_px", "traj_cw"):
                continue
            land = synth.synth_landscape(case["landscape"]["seed"], case["landscape"]["size"], case["landscape"]["grain"])
            model = sharded.device_sharded_sads_familiarity(case["chem_weight"], 0, 1, "cuda:0")
            nsf, best, pos, ang, fam, status = _run_trajectory(case, land, model, track_scene_familiarity=False)
            name = case["name"]
            assert isinstance(nsf._engine, sharded.ShardedDeviceEngine) and nsf._engine.exchange.exchanges >= len(best)
            assert status == case["stop_status"] and np.array_equal(best, z[name + "_best"])
            assert pos.tobytes() == z[name + "_pos"].tobytes() and ang.tobytes() == z[name + "_angle"].tobytes()
            np.testing.assert_allclose(fam, z[name + "_fam"], rtol=1e-12, atol=0)
            assert hashlib.sha256(np.ascontiguousarray(nsf.familiar_scenes).tobytes()).digest() == bytes(z[name + "_scenes_sha"])
            nsf._engine.close()
        # the sensor running off the landscape inside a sharded step: the record carries it, every rank raises the
        # reference's IndexError (the single-GPU agent does: test_gpu_sensor_index_errors_like_the_reference)
        land = synth.synth_landscape(3, 120, 4)
        model = sharded.device_sharded_sads_familiarity(0.25, 0, 1, "cuda:0")
        nsf = navsim_amd.NavBySceneFamiliarity(land, (40, 40), 1.0, n_test_angles=4, familiarity_model=model,
                                               track_scene_familiarity=False)
        nsf.train_from_path(np.stack([np.linspace(40, 80, 30), np.full(30, 60.0)], axis=1))
        nsf.position, nsf.angle = (99.4, 99.4), 0.8 - nsf.angle_offsets[0]
        with pytest.raises(IndexError):
            nsf.step_forward(fake=True)
        nsf.position, nsf.angle = (60.0, 60.0), 0.0
        nsf.step_forward(fake=True)                      # and the exchange is usable afterwards
        nsf._engine.close()
    finally:
        dist.destroy_process_group()


def _mailbox_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)      # rendezvous only: the exchange is the mailbox
    out = []
    try:
        eng = navsim_amd.FamiliarityEngine(0)
        for name, cw in (("plain", 0.0), ("ties", 0.3)):
            lib, pats = _library_with_ties()
            if name == "plain":
                pats = synth.synth_patches(78, 7, 12, 12)
                pats[2] = synth.near_match_patch(lib[700], 3)
            lo, hi = sharded.shard_bounds(len(lib), world, rank)
            eng.set_library(lib[lo:hi], cw, first_view=lo)
            eng.upload_patches(pats)
            ex = sharded.MailboxExchange(eng, rank, world)
            for _ in range(5):                                         # slots are reused from the third step on
                res = ex.step()
            out.append((name, res["best_idex"], res["best_view"], float(res["step_familiarity"]),
                        np.asarray(res["angle_familiarity"]).tolist(), ex.exchanges))
            ex.close()
        eng.close()
        q.put((rank, out))
    finally:
        dist.barrier()
        dist.destroy_process_group()


def test_mailbox_exchange_three_ranks_on_one_gpu():
    """The mailbox exchange with three processes sharing this GPU (one context and one library shard each): decisions,
    cross-rank ties included, equal the unsharded reference on every rank."""
    import torch.multiprocessing as mp
    world = 3
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_mailbox_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    lib, tie_pats = _library_with_ties()
    plain = synth.synth_patches(78, 7, 12, 12)
    plain[2] = synth.near_match_patch(lib[700], 3)
    want = {"plain": oracle.step(lib, plain, 0.0), "ties": oracle.step(lib, tie_pats, 0.3)}
    for rank in range(world):
        for name, best, view, fam, angle, exchanges in got[rank]:
            w = want[name]
            assert (best, view) == (w["best_idex"], w["best_view"]), (rank, name)
            np.testing.assert_allclose(fam, w["step_familiarity"], rtol=1e-12)
            np.testing.assert_allclose(angle, w["angle_familiarity"], rtol=1e-12)
            assert exchanges >= (10 if name == "ties" else 5)          # ties take the second round every step
