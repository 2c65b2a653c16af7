"""View-library sharding across the GPUs of a node: one process and one engine per GPU.

The stored views are independent units (the reference's kernel loops `for fam_idex`,
navsim/util.pyx:44, and a step only needs the per-heading maximum over views,
navsim/NavBySceneFamiliarity.py:313), so rank r keeps the contiguous block
[r*F/N, (r+1)*F/N) of the library and every rank scores the same patches.  The only exchange
per step is one all-gather of a (3+4A)-double record per rank over RCCL (backend "nccl" on
ROCm; xGMI is point-to-point and the message is a few hundred bytes, so this is one
latency-bound hop), after which every rank reduces the records identically:

    angle_familiarity[a] = max_r angle_fam_r[a]
    best heading         = np.argmax rule on exact values whenever the integer scores tie

Tie protocol (keeps `best_idex` bit-identical to the unsharded reference): each rank reports the
number of (heading, view) pairs within delta of ITS OWN maximum.  A rank whose maximum is within
delta of the global maximum is "contending"; if the contending ranks hold more than one
candidate in total, they re-score their candidates with the exact sequential-double kernel
(`resolve`) and a second all-gather carries the exact per-heading maxima.
"""
import ctypes
import os

import numpy as np


CANDIDATE_CAP = 4096        # kCandCap of csrc/dejavu_kernels.h


def shard_bounds(n_views, world_size, rank):
    """Contiguous block of views owned by `rank` (np.array_split convention: first ranks get the extras)."""
    base, extra = divmod(int(n_views), int(world_size))
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def pack_record(res):
    """Per-rank record exchanged each step: [approx_max, n_candidates, resolved, angle_fam[A], angle_view[A], exact_fam[A], exact_view[A]]."""
    A = len(res["angle_familiarity"])
    rec = np.empty(3 + 4 * A, dtype=np.float64)
    rec[0] = res["approx_max"]
    rec[1] = float(res["n_candidates"])
    # 0: integer-sum scores only; 1: candidates re-scored exactly; 2: every score exact
    rec[2] = 2.0 if (res["flags"] & 2) else (1.0 if (res["flags"] & 1) else 0.0)
    rec[3:3 + A] = res["angle_familiarity"]
    rec[3 + A:3 + 2 * A] = res["angle_view"]
    rec[3 + 2 * A:3 + 3 * A] = res["exact_familiarity"]
    rec[3 + 3 * A:3 + 4 * A] = res["exact_view"]
    return rec


SENSE_ERROR_MESSAGE = "sensor footprint reaches past the end of the landscape (index out of bounds)"


def check_sense_error(records):
    """A rank whose patches were sensed past the end of the landscape says so in its record's state word (+ 4): every
    rank raises the reference's IndexError (util.pyx:137-168) instead of deciding on zeroed pixels."""
    if np.any(records[:, 2] >= 4.0):
        raise IndexError(SENSE_ERROR_MESSAGE)


def contending(records, delta):
    """Indices of the ranks whose local maximum is within delta of the global maximum."""
    check_sense_error(records)
    gmax = np.max(records[:, 0])
    return [r for r in range(records.shape[0]) if records[r, 0] >= gmax - delta], gmax


def needs_resolve(records, delta):
    ranks, _ = contending(records, delta)
    total = sum(int(records[r, 1]) for r in ranks)
    unresolved = [r for r in ranks if records[r, 2] == 0.0]
    return total > 1 and len(unresolved) > 0, ranks


def merge_records(records, delta, n_headings):
    """Global decision from the gathered records; identical on every rank.

    Returns dict(best_idex, best_view, step_familiarity, angle_familiarity[A]).
    """
    A = n_headings
    ranks, _ = contending(records, delta)
    total = sum(int(records[r, 1]) for r in ranks)
    ang = records[:, 3:3 + A]
    view = records[:, 3 + A:3 + 2 * A]
    angle_fam = ang.max(axis=0)
    if total <= 1:
        # a single candidate pair in the whole library: the integer scores decide
        best = int(np.argmax(angle_fam))
        owner = int(np.argmax(ang[:, best]))
        return dict(best_idex=best, best_view=int(view[owner, best]), step_familiarity=float(angle_fam[best]),
                    angle_familiarity=angle_fam, resolved=False)
    exact = np.full((records.shape[0], A), -np.inf)
    exview = np.full((records.shape[0], A), -1.0)
    for r in ranks:
        if records[r, 2] == 0.0:
            raise RuntimeError("rank %d contends for the maximum but did not resolve its candidates" % r)
        if records[r, 2] == 2.0:                      # exact everywhere: the per-heading maxima are exact
            exact[r], exview[r] = ang[r], view[r]
        else:
            exact[r] = records[r, 3 + 2 * A:3 + 3 * A]
            exview[r] = records[r, 3 + 3 * A:3 + 4 * A]
    ex_a = exact.max(axis=0)
    best = int(np.argmax(ex_a))                      # first maximum, NavBySceneFamiliarity.py:315
    owners = [r for r in ranks if exact[r, best] == ex_a[best]]
    best_view = int(min(exview[r, best] for r in owners))
    out_fam = np.where(np.isfinite(ex_a), ex_a, angle_fam)
    return dict(best_idex=best, best_view=best_view, step_familiarity=float(ex_a[best]),
                angle_familiarity=out_fam, resolved=True)


# ---- the fast exchange: ONE all-reduce(max) of packed uint64 keys per step ---------------------------------------
# (the north star's collective; include/dejavu.h:dv_step_keys describes the words).  It decides the step whenever a
# single (heading, view) pair lies within delta of the global maximum -- the usual case; near-ties fall back to the
# exchange of full records above.  pack_keys / merge_keys are the readable statement of k_make_keys / dv_merge_keys.
KEY_WORDS_PER_RANK = 4
_TOP = np.uint64(1 << 63)


def ordered_key(x):
    """uint64 image of float64 whose unsigned order is the numeric order (csrc/dejavu_kernels.h:ordered_key)."""
    b = np.ascontiguousarray(x, dtype=np.float64).view(np.uint64)
    return np.where(b >> np.uint64(63), ~b, b | _TOP)


def key_to_double(k):
    k = np.ascontiguousarray(k, dtype=np.uint64)
    b = np.where(k >> np.uint64(63), k & ~_TOP, ~k)
    return b.view(np.float64)


def pack_keys(rec, rank, world):
    """This rank's words of the key exchange from its packed record (pack_record): uint64[A + 4 * world]."""
    A = (len(rec) - 3) // 4
    keys = np.zeros(A + KEY_WORDS_PER_RANK * world, dtype=np.uint64)
    ang = np.asarray(rec[3:3 + A], dtype=np.float64)
    keys[:A] = ordered_key(ang)
    best = int(np.argmax(ang))                                           # first maximum
    base = A + KEY_WORDS_PER_RANK * rank
    keys[base] = ordered_key(np.array([rec[0]]))[0]
    keys[base + 1] = np.uint64(min(int(rec[1]), 0x7fffffff) | (int(rec[2]) << 32) | (1 << 48))
    keys[base + 2] = np.uint64(best + 1)
    keys[base + 3] = np.uint64(int(rec[3 + A + best]) + 1)
    return keys


def merge_keys(keys, world, n_headings, delta):
    """Decision from the max-reduced keys, or None when near-ties need the exchange of full records."""
    A = n_headings
    keys = np.ascontiguousarray(keys, dtype=np.uint64)
    slots = keys[A:A + KEY_WORDS_PER_RANK * world].reshape(world, KEY_WORDS_PER_RANK)
    if np.any((slots[:, 1] >> np.uint64(48)) == 0):
        raise RuntimeError("a rank contributed no slot to the key exchange")
    if np.any(((slots[:, 1] >> np.uint64(32)) & np.uint64(0xff)) >= 4):
        raise IndexError(SENSE_ERROR_MESSAGE)
    approx = key_to_double(slots[:, 0])
    gmax = approx.max()
    total = int(np.sum((slots[:, 1] & np.uint64(0xffffffff))[approx >= gmax - delta]))
    if total > 1:
        return None
    winner = int(np.argmax(approx))
    angle_fam = key_to_double(keys[:A]).copy()
    best = int(slots[winner, 2]) - 1
    return dict(best_idex=best, best_view=int(slots[winner, 3]) - 1, step_familiarity=float(angle_fam[best]),
                angle_familiarity=angle_fam, resolved=False)


def merge_keys_native(keys, world, n_headings, delta, signed_order=False):
    """merge_keys through the library (dv_merge_keys, host arithmetic only)."""
    from . import _native as N
    lib = N.load()
    keys = np.ascontiguousarray(keys).view(np.uint64)
    out = N.MergeOut()
    rc = lib.dv_merge_keys(keys.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)), int(world), int(n_headings), float(delta),
                           1 if signed_order else 0, out)
    if rc == -5:
        raise IndexError(SENSE_ERROR_MESSAGE)
    if rc != 0:
        raise N.EngineError("dv_merge_keys failed (%d)" % rc)
    if out.needs_resolve:
        return None
    fam = np.frombuffer(out, dtype=np.float64, count=int(n_headings), offset=N.MergeOut.angle_fam.offset)
    return dict(best_idex=out.best_heading, best_view=out.best_view, step_familiarity=out.best_fam,
                angle_familiarity=fam, resolved=False)


def merge_records_native(records, delta, n_headings):
    """needs_resolve + merge_records in one call into the library (dv_merge_records, host arithmetic only).

    Returns (again, ranks, decision): `again`/`ranks` as needs_resolve, `decision` as merge_records (None when
    `again`).  The Python functions above are the readable statement of the same rules; tests compare the two.
    """
    from . import _native as N
    lib = N.load()
    records = np.ascontiguousarray(records, dtype=np.float64)
    world, stride = records.shape
    out = N.MergeOut()
    rc = lib.dv_merge_records(N.f64ptr(records), world, int(n_headings), stride, float(delta), out)
    if rc == -5:                                         # DV_ERR_INDEX: some rank sensed past the end of the landscape
        raise IndexError(SENSE_ERROR_MESSAGE)
    if rc != 0:
        raise N.EngineError("dv_merge_records failed (%d)" % rc)
    mask = out.contending_mask
    ranks = [r for r in range(world) if (mask >> r) & 1]
    if out.needs_resolve:
        return True, ranks, None
    fam = np.frombuffer(out, dtype=np.float64, count=int(n_headings), offset=N.MergeOut.angle_fam.offset)
    return False, ranks, dict(best_idex=out.best_heading, best_view=out.best_view, step_familiarity=out.best_fam,
                              angle_familiarity=fam, resolved=bool(out.resolved))


class ShardedFamiliarity(object):
    """One rank's share of a sharded library plus the per-step exchange.

    `engine`  : FamiliarityEngine (or any object with step/resolve returning the same dicts)
    `gather`  : callable(np.ndarray[k]) -> np.ndarray[world, k], the all-gather (torch.distributed
                in production, see `torch_gather`; tests inject a gloo one)
    """

    def __init__(self, engine, gather, rank, world_size, reduce_max=None):
        self.engine = engine
        self.gather = gather
        self.reduce_max = reduce_max          # callable(uint64[k]) -> elementwise maximum over the ranks, or None
        self.rank = rank
        self.world_size = world_size
        self.exchanges = 0
        self.key_decisions = 0                # steps decided by the all-reduce(max) of packed keys alone
        self.n_views_total = None
        self._scene_local = None              # this rank's block of the last step's per-view minimum (want_scene=True)

    def set_library(self, scenes, chem_weight=0.0):
        """Every rank passes the FULL library (or only its own block via set_library_block)."""
        lo, hi = shard_bounds(len(scenes), self.world_size, self.rank)
        self.engine.set_library(scenes[lo:hi], chem_weight, first_view=lo)
        self.bounds = (lo, hi)
        self.n_views_total = len(scenes)
        self._scene_local = None

    def gather_scene_familiarity(self):
        """scene_familiarity[F] of the last step (navsim/NavBySceneFamiliarity.py:301-303), gathered from the ranks' blocks.

        The per-view minimum over the headings is per view, so a step leaves it sharded (`scene_familiarity_local`) and pays
        nothing for it; the reference only reads it for plots (:540,630).  A COLLECTIVE: every rank calls it (one all-gather of
        ceil(F / world) doubles per rank), after a step taken with want_scene=True."""
        if self._scene_local is None:
            raise RuntimeError("no per-view minimum to gather: the last step was not taken with want_scene=True")
        per = (self.n_views_total + self.world_size - 1) // self.world_size
        mine = np.full(per, np.nan)
        mine[:len(self._scene_local)] = self._scene_local
        allv = np.asarray(self.gather(mine)).reshape(self.world_size, per)
        out = np.empty(self.n_views_total, dtype=np.float64)
        for r in range(self.world_size):
            lo, hi = shard_bounds(self.n_views_total, self.world_size, r)
            out[lo:hi] = allv[r, :hi - lo]
        return out

    def step(self, patches, want_scene=False):
        res = self.engine.step(patches, want_scene=want_scene)
        self._scene_local = res.get("scene_familiarity") if want_scene else None
        A = len(res["angle_familiarity"])
        delta = res["delta"]
        if self.reduce_max is not None:
            # one all-reduce(max) of A + 4*world words decides unless there are near-ties
            keys = self.reduce_max(pack_keys(pack_record(res), self.rank, self.world_size))
            self.exchanges += 1
            out = merge_keys(keys, self.world_size, A, delta)
            if out is not None:
                self.key_decisions += 1
                out["scene_familiarity_local"] = self._scene_local
                out["scene_familiarity"] = None              # sharded: gather_scene_familiarity() when it is read
                return out
        records = self.gather(pack_record(res))
        self.exchanges += 1
        again, ranks = needs_resolve(records, delta)
        if again:
            if self.rank in ranks and not (res["flags"] & 3):
                res = self.engine.resolve()
            records = self.gather(pack_record(res))
            self.exchanges += 1
        out = merge_records(records, delta, A)
        out["scene_familiarity_local"] = self._scene_local
        out["scene_familiarity"] = None                      # sharded: gather_scene_familiarity() when it is read
        return out


def sharded_sads_familiarity(chem_weight, gather, rank, world_size, engine_factory=None, reduce_max=None):
    """The reference's plug-in shape (navsim/util.pyx:10-25) over a library sharded across ranks.

    model(scenes) gives this rank's engine its block of `scenes` and returns `func` whose `.engine` is a
    ShardedFamiliarity: navsim_amd.NavBySceneFamiliarity then runs its fused step through the exchange, and every
    rank takes the same decision, so the same agent code runs unchanged on every rank (construct the agent with
    use_gpu_sensor=False, track_scene_familiarity=False: the per-view minimum stays sharded).  Calling `func`
    itself (one heading, all views) is not offered: the per-view scores live on different ranks.
    """
    def model(scenes):
        assert 0 <= chem_weight <= 1
        if engine_factory is not None:
            engine = engine_factory()
        else:
            from .engine import FamiliarityEngine
            engine = FamiliarityEngine()
        sh = ShardedFamiliarity(engine, gather, rank, world_size, reduce_max=reduce_max)
        sh.set_library(scenes, chem_weight)

        def func(scene, fambuf):
            raise NotImplementedError("per-view scores of a sharded library live on different ranks; use the agent's "
                                      "fused step (func.engine.step)")

        func.max_familiarity = scenes[0].shape[0] * scenes[0].shape[1]
        func.engine = sh
        func.chem_weight = chem_weight
        return func

    model.chem_weight = chem_weight
    return model


class ShardedEnsemble(object):
    """Ensemble of independent agents (BASELINE.json configs[4]; the reference farms trials over MPI ranks,
    scripts/run_experiment.py:326-347): the AGENTS are partitioned over the ranks in contiguous blocks and every
    rank holds the whole library, so a step has no data-path collective at all.  `decisions` gathers the three
    numbers per agent that a driver logs; it is reporting, not part of the step.
    """

    def __init__(self, engine, rank, world_size):
        self.engine = engine
        self.rank = rank
        self.world_size = world_size

    def set_library(self, scenes, chem_weight=0.0):
        self.engine.set_library(scenes, chem_weight)

    def agent_bounds(self, n_agents):
        return shard_bounds(n_agents, self.world_size, self.rank)

    def step(self, patches):
        """patches uint8[N, A, h, w, 3] for ALL agents (or only this rank's block with `local=True` semantics:
        pass patches[lo:hi] and use step_local).  Returns this rank's agents' result dicts, in agent order."""
        lo, hi = self.agent_bounds(len(patches))
        return self.step_local(patches[lo:hi])

    def step_local(self, patches_local):
        if len(patches_local) == 0:
            return []
        return self.engine.step_batch(np.ascontiguousarray(patches_local))

    def decisions(self, results, n_agents, gather):
        """All ranks' (best_idex, best_view, step_familiarity) as float64[n_agents, 3] via one all-gather."""
        per = (n_agents + self.world_size - 1) // self.world_size
        mine = np.full((per, 3), np.nan)
        for i, r in enumerate(results):
            mine[i] = (r["best_idex"], r["best_view"], r["step_familiarity"])
        allr = gather(mine.reshape(-1)).reshape(self.world_size, per, 3)
        out = np.empty((n_agents, 3))
        for rk in range(self.world_size):
            lo, hi = shard_bounds(n_agents, self.world_size, rk)
            out[lo:hi] = allr[rk, :hi - lo]
        return out


def torch_gather(device=None):
    """All-gather over torch.distributed (RCCL when the tensors live on the GPU, gloo on CPU)."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size()

    def gather(vec):
        t = torch.from_numpy(np.ascontiguousarray(vec, dtype=np.float64))
        if device is not None:
            t = t.to(device)
        out = torch.empty(world * t.numel(), dtype=t.dtype, device=t.device)
        dist.all_gather_into_tensor(out, t.reshape(-1))
        return out.cpu().numpy().reshape((world,) + tuple(t.shape))

    return gather


def torch_reduce_max(device=None):
    """reduce_max for ShardedFamiliarity over an initialised torch.distributed group: elementwise maximum of uint64
    words over the ranks.  torch reduces int64, so the words travel with their top bit flipped (signed order = unsigned
    order)."""
    import torch
    import torch.distributed as dist

    def reduce_max(keys):
        flipped = (np.ascontiguousarray(keys, dtype=np.uint64) ^ _TOP).view(np.int64)
        t = torch.from_numpy(flipped.copy())
        if device is not None:
            t = t.to(device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return t.cpu().numpy().view(np.uint64) ^ _TOP

    return reduce_max


class _NcclUniqueId(ctypes.Structure):
    _fields_ = [("internal", ctypes.c_char * 128)]


class DirectRccl(object):
    """RCCL all-gather issued straight onto the engine's stream (ctypes into the librccl.so torch ships).

    torch.distributed runs its collectives on a side stream and ties it to the caller's stream with two event
    waits; for a 2 KB record behind a ~110 us step those hops are a visible share.  The communicator is created from
    an ncclUniqueId that rank 0 makes and torch.distributed broadcasts; everything else (rendezvous, barriers,
    timing) stays with torch.distributed.  Raises on any failure so that the caller can keep torch's path.
    """

    NCCL_FLOAT64 = 8
    NCCL_UINT64 = 5
    NCCL_MAX = 2

    def __init__(self, rank, world_size):
        import torch
        import torch.distributed as dist
        path = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
        self.lib = ctypes.CDLL(path)
        self.lib.ncclGetErrorString.restype = ctypes.c_char_p
        self.lib.ncclCommInitRank.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, _NcclUniqueId, ctypes.c_int]
        self.lib.ncclAllGather.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int,
                                           ctypes.c_void_p, ctypes.c_void_p]
        self.lib.ncclAllReduce.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int,
                                           ctypes.c_void_p, ctypes.c_void_p]
        uid = _NcclUniqueId()
        if rank == 0:
            self._ok(self.lib.ncclGetUniqueId(ctypes.byref(uid)), "ncclGetUniqueId")
        # raw bytes (a c_char array read through .value would stop at the first NUL)
        box = [ctypes.string_at(ctypes.addressof(uid), 128) if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        ctypes.memmove(ctypes.addressof(uid), box[0], 128)
        self.comm = ctypes.c_void_p()
        self._ok(self.lib.ncclCommInitRank(ctypes.byref(self.comm), int(world_size), uid, int(rank)), "ncclCommInitRank")

    def _ok(self, rc, what):
        if rc != 0:
            raise RuntimeError("%s failed: %s" % (what, self.lib.ncclGetErrorString(rc).decode()))

    def all_gather_f64(self, send_ptr, recv_ptr, count, stream):
        self._ok(self.lib.ncclAllGather(ctypes.c_void_p(send_ptr), ctypes.c_void_p(recv_ptr), count, self.NCCL_FLOAT64,
                                        self.comm, ctypes.c_void_p(stream)), "ncclAllGather")

    def all_reduce_max_u64(self, send_ptr, recv_ptr, count, stream):
        self._ok(self.lib.ncclAllReduce(ctypes.c_void_p(send_ptr), ctypes.c_void_p(recv_ptr), count, self.NCCL_UINT64,
                                        self.NCCL_MAX, self.comm, ctypes.c_void_p(stream)), "ncclAllReduce")

    def comm_count(self):
        """ncclCommCount of the communicator the per-step collective runs on: the number of ranks RCCL itself sees."""
        n = ctypes.c_int(-1)
        self._ok(self.lib.ncclCommCount(self.comm, ctypes.byref(n)), "ncclCommCount")
        return int(n.value)

    def close(self):
        if self.comm:
            self.lib.ncclCommDestroy(self.comm)
            self.comm = ctypes.c_void_p()


class _DeviceArray(object):
    """Minimal __cuda_array_interface__ carrier so that torch can wrap a raw device pointer without copying."""

    def __init__(self, ptr, n, typestr="<f8"):
        self.__cuda_array_interface__ = dict(shape=(n,), typestr=typestr, data=(ptr, False), version=2)


class DeviceExchange(object):
    """Per-step exchange that never leaves the GPU until the gathered records are complete.

    The engine's kernels run on one stream and leave this rank's packed record in device memory (dv_step_record);
    the RCCL all-gather is ordered behind them on the same stream, then the engine's hand-over kernel (dv_publish)
    copies all ranks' records to mapped host memory, and the host polls ONE sequence word per exchange.  Cross-rank near-ties take a
    second round (dv_resolve_enqueue + all-gather), exactly like ShardedFamiliarity.step.
    """

    def __init__(self, engine, rank, world_size, device):
        import torch
        import torch.distributed as dist
        self._torch, self._dist = torch, dist
        self.engine, self.rank, self.world = engine, rank, world_size
        self.device = torch.device(device)
        # A stream of its own, shared by the engine's kernels and the collective.  (torch's default stream has the null
        # handle, which dv_set_stream reads as "the context's own stream" -- a non-blocking one the null stream does not
        # order with: the all-gather could then read the record before k_tail has written it.)
        self.stream = torch.cuda.Stream(device=self.device)
        assert self.stream.cuda_stream != 0
        engine.set_stream(self.stream.cuda_stream)
        self.delta = engine.library_info()["delta"]
        ptr, n = engine.step_record()
        self.n = n
        self.A = (n - 3) // 4
        self.record = torch.as_tensor(_DeviceArray(ptr, n), device=self.device)
        self.gathered = torch.empty(world_size * n, dtype=torch.float64, device=self.device)
        self.host = np.empty((world_size, n), dtype=np.float64)
        self.exchanges = 0
        # the fast exchange: one all-reduce(max) of A + 4*world packed keys (DEJAVU_KEY_EXCHANGE=0: always full records)
        self.use_keys = os.environ.get("DEJAVU_KEY_EXCHANGE", "1") != "0"
        self.n_keys = self.A + KEY_WORDS_PER_RANK * world_size
        self.keys_red = torch.zeros(self.n_keys, dtype=torch.int64, device=self.device)
        self.keys_host = np.empty(self.n_keys, dtype=np.float64)      # raw words; dv_publish moves 8-byte units
        self.key_decisions = 0
        # RCCL on the engine's own stream when it can be set up on every rank (DEJAVU_DIRECT_RCCL=0: torch's path)
        self.direct = None
        if os.environ.get("DEJAVU_DIRECT_RCCL", "1") != "0":
            torch.cuda.set_device(self.device)           # RCCL binds the communicator to the calling thread's device
            try:
                direct = DirectRccl(rank, world_size)
                ok = 1
            except Exception as e:                       # noqa: BLE001 - any failure means: use torch's collective
                direct, ok = None, 0
                self.direct_error = repr(e)
            flag = torch.tensor([ok], dtype=torch.int32, device=self.device)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if int(flag.item()) == 1:
                self.direct = direct
                self._self_test()
            elif direct is not None:
                direct.close()

    def _self_test(self):
        """One all-gather of the rank numbers through the direct communicator; falls back to torch's path on mismatch."""
        torch = self._torch
        send = torch.full((self.n,), float(self.rank), dtype=torch.float64, device=self.device)
        torch.cuda.synchronize(self.device)              # `send` was filled on torch's current stream
        self.direct.all_gather_f64(send.data_ptr(), self.gathered.data_ptr(), self.n, self.stream.cuda_stream)
        self.stream.synchronize()
        got = self.gathered.view(self.world, self.n)[:, 0].cpu().numpy()
        ok = int(np.array_equal(got, np.arange(self.world, dtype=np.float64)))
        flag = torch.tensor([ok], dtype=torch.int32, device=self.device)
        self._dist.all_reduce(flag, op=self._dist.ReduceOp.MIN)
        if int(flag.item()) != 1:
            self.direct.close()
            self.direct = None

    def close(self):
        self.stream.synchronize()
        if self.direct is not None:
            self.direct.close()
            self.direct = None
        self.engine.set_stream(None)                     # back to the context's own stream

    def _gather(self):
        # all-gather (RCCL) behind the step's kernels on the same stream, then the engine's own hand-over kernel:
        # the host polls one sequence word instead of blocking on the stream
        if self.direct is not None:
            self.direct.all_gather_f64(self.record.data_ptr(), self.gathered.data_ptr(), self.n, self.stream.cuda_stream)
        else:
            with self._torch.cuda.stream(self.stream):
                self._dist.all_gather_into_tensor(self.gathered, self.record)
        self.engine.publish(self.gathered.data_ptr(), self.world * self.n)
        self.engine.publish_wait(self.host)
        self.exchanges += 1
        return self.host

    def _reduce_keys(self):
        """One all-reduce(max) of this step's packed keys behind the step's kernels, handed to the host like the
        records; returns the decision or None (near-ties: the full records decide)."""
        direct = self.direct is not None
        ptr, n = self.engine.step_keys(self.rank, self.world, signed_order=not direct)
        assert n == self.n_keys
        if direct:
            self.direct.all_reduce_max_u64(ptr, self.keys_red.data_ptr(), n, self.stream.cuda_stream)
        else:
            with self._torch.cuda.stream(self.stream):
                src = self._torch.as_tensor(_DeviceArray(ptr, n, "<i8"), device=self.device)
                self.keys_red.copy_(src)
                self._dist.all_reduce(self.keys_red, op=self._dist.ReduceOp.MAX)
        self.engine.publish(self.keys_red.data_ptr(), n)
        self.engine.publish_wait(self.keys_host)
        self.exchanges += 1
        return merge_keys_native(self.keys_host, self.world, self.A, self.delta, signed_order=not direct)

    def rccl_ranks(self):
        """Ranks of the RCCL communicator the exchange runs on (ncclCommCount), or torch.distributed's world size when the
        collective goes through torch."""
        if self.direct is not None:
            return self.direct.comm_count()
        return int(self._dist.get_world_size())

    def exchange_only_us(self, n=50):
        """Microseconds per exchange with nothing else on the stream: the last step's keys reduced again n times (one
        all-reduce(max) + hand-over to the host each).  What a step pays for being sharded, beside its scoring."""
        import time
        if not self.use_keys:
            return None
        self._reduce_keys()
        t0 = time.perf_counter()
        for _ in range(n):
            self._reduce_keys()
        return (time.perf_counter() - t0) / n * 1e6

    def step(self):
        self.engine.step_enqueue(want_scene=False)
        lib = self.engine._lib
        lib.dv_range_push(b"dv:exchange")                # roctx range (no-op unless DEJAVU_ROCTX=1)
        try:
            return self._exchange()
        finally:
            lib.dv_range_pop()

    def _exchange(self):
        if self.use_keys:
            out = self._reduce_keys()
            if out is not None:
                self.key_decisions += 1
                return out
        records = self._gather()
        again, ranks, out = merge_records_native(records, self.delta, self.A)
        if again:
            if self.rank in ranks and records[self.rank, 2] == 0.0:
                if records[self.rank, 1] > CANDIDATE_CAP:
                    # more local near-ties than the candidate list holds: score this shard exactly instead
                    self.engine.set_exact(True)
                    self.engine.step_enqueue(want_scene=False)
                    self.engine.set_exact(False)
                else:
                    self.engine.resolve_enqueue()
            records = self._gather()
            again, ranks, out = merge_records_native(records, self.delta, self.A)
            if again:
                raise RuntimeError("contending ranks %r did not resolve their candidates" % (ranks,))
        return out


class MailboxExchange(object):
    """Per-step exchange of the ranks of ONE node through a host segment they all map (dv_set_mailbox): every rank's
    GPU writes its 2 KB record into its entry of the segment behind the step's kernels (one tiny kernel, the stores
    cross PCIe), every rank's host polls the entries.  No collective kernel, no stream hops, no device-to-host copy;
    torch.distributed (any backend) is used once, to agree on the segment's name.  Same records, same merge and same
    tie protocol as DeviceExchange, hence the same decisions.  Opt-in: bench.py --exchange mailbox.

    Slots: (step parity) x (round 1 / round 2).  Reusing a slot two steps later is safe: nobody posts step s + 2 before
    everybody has posted step s + 1, which each rank does only after it has finished reading step s.
    """

    SLOTS = 4
    ENTRY = 512                     # doubles per (slot, rank) entry, kMboxEntry of csrc/dejavu_kernels.h

    def __init__(self, engine, rank, world_size, rendezvous=None, timeout_ms=20000):
        import mmap
        self.engine, self.rank, self.world, self.timeout_ms = engine, rank, world_size, timeout_ms
        if rendezvous is None:
            import torch.distributed as dist

            def rendezvous(obj):
                box = [obj]
                dist.broadcast_object_list(box, src=0)
                dist.barrier()
                return box[0]
        size = (self.SLOTS * world_size * self.ENTRY * 8 + 4095) // 4096 * 4096
        name = None
        if rank == 0:
            name = "/dev/shm/dejavu_mbox_%d_%08x" % (os.getpid(), int.from_bytes(os.urandom(4), "little"))
            with open(name, "wb") as f:
                f.truncate(size)
        name = rendezvous(name)                              # everybody knows the name, and the file exists
        fd = os.open(name, os.O_RDWR)
        try:
            self._mm = mmap.mmap(fd, size)
        finally:
            os.close(fd)
        rendezvous(None)                                     # everybody has mapped it: the name can go
        if rank == 0:
            os.unlink(name)
        self._anchor = ctypes.c_char.from_buffer(self._mm)
        engine.set_mailbox(ctypes.addressof(self._anchor), size, rank, world_size)
        self.delta = engine.library_info()["delta"]
        _, n = engine.step_record()
        self.n, self.A = n, (n - 3) // 4
        self.host = np.empty((world_size, n), dtype=np.float64)
        self._round2 = np.empty((world_size, n), dtype=np.float64)
        self.all_mask = (1 << world_size) - 1
        self.step_no = 0
        self.exchanges = 0

    def step(self):
        self.step_no += 1
        base = (self.step_no & 1) * 2
        seq = 2 * self.step_no
        eng = self.engine
        eng.step_enqueue(want_scene=False)
        eng.mailbox_post(base, seq)
        records = eng.mailbox_wait(base, seq, self.all_mask, self.host, self.timeout_ms)
        self.exchanges += 1
        again, ranks, out = merge_records_native(records, self.delta, self.A)
        if again:
            redo = [r for r in ranks if records[r, 2] == 0.0]        # known to every rank from the first round
            if self.rank in redo:
                if records[self.rank, 1] > CANDIDATE_CAP:
                    eng.set_exact(True)
                    eng.step_enqueue(want_scene=False)
                    eng.set_exact(False)
                else:
                    eng.resolve_enqueue()
                eng.mailbox_post(base + 1, seq + 1)
            mask = 0
            for r in redo:
                mask |= 1 << r
            eng.mailbox_wait(base + 1, seq + 1, mask, self._round2, self.timeout_ms)
            for r in redo:
                records[r] = self._round2[r]
            self.exchanges += 1
            again, ranks, out = merge_records_native(records, self.delta, self.A)
            if again:
                raise RuntimeError("contending ranks %r did not resolve their candidates" % (ranks,))
        return out

    def close(self):
        self.engine.synchronize()
        self.engine.set_mailbox(0, 0, 0, 1)
        del self._anchor
        self._mm.close()


class ShardedDeviceEngine(object):
    """What navsim_amd.NavBySceneFamiliarity needs from its engine, over a library sharded across ranks with the
    sensor model on every GPU: the landscape is replicated, each rank senses and keeps its own block of the training
    views, and a step senses the heading patches locally (same bytes on every rank), scores them against the local
    block and takes the global decision through DeviceExchange.  One rank = one process = one GPU.
    """

    senses_on_device = True

    def __init__(self, engine, rank, world_size, device, gather_views=True):
        self.engine, self.rank, self.world, self.device = engine, rank, world_size, device
        self.gather_views = gather_views
        self.exchange = None
        self.n_views = 0

    # -- passed through to this rank's context
    def set_landscape(self, landscape):
        self.engine.set_landscape(landscape)

    def configure_sensor(self, *args, **kwargs):
        self.engine.configure_sensor(*args, **kwargs)

    def sense(self, x, y, angle):
        return self.engine.sense(x, y, angle)

    def set_library_from_poses(self, x, y, angle, chem_weight=0.0):
        """Every rank passes the whole path; it senses and ingests views [lo, hi) and returns uint8[F,h,w,3] with the
        other ranks' blocks filled in by one all-gather (gather_views) or left zero."""
        x, y, angle = (np.ascontiguousarray(v, dtype=np.float64) for v in (x, y, angle))
        F = len(x)
        lo, hi = shard_bounds(F, self.world, self.rank)
        mine = self.engine.set_library_from_poses(x[lo:hi], y[lo:hi], angle[lo:hi], chem_weight, first_view=lo)
        self.bounds, self.n_views = (lo, hi), F
        views = np.zeros((F,) + mine.shape[1:], dtype=np.uint8)
        views[lo:hi] = mine
        if self.gather_views and self.world > 1:
            import torch
            import torch.distributed as dist
            per = (F + self.world - 1) // self.world
            block = int(np.prod(mine.shape[1:]))
            send = torch.zeros(per * block, dtype=torch.uint8, device=self.device)
            send[:(hi - lo) * block] = torch.from_numpy(mine.reshape(-1)).to(self.device)
            recv = torch.empty(self.world * per * block, dtype=torch.uint8, device=self.device)
            dist.all_gather_into_tensor(recv, send)
            allv = recv.cpu().numpy().reshape(self.world, per, *mine.shape[1:])
            for r in range(self.world):
                rlo, rhi = shard_bounds(F, self.world, r)
                views[rlo:rhi] = allv[r, :rhi - rlo]
        if self.exchange is not None:
            self.exchange.close()
            self.exchange = None
        return views

    def sense_step(self, x, y, angles, want_scene=False, force_resolve=False):
        if want_scene:
            raise ValueError("scene_familiarity stays sharded: construct the agent with track_scene_familiarity=False")
        self.engine.sense_patches(x, y, angles)
        if self.exchange is None:                        # needs resident patches (record size) and the library (delta)
            self.exchange = DeviceExchange(self.engine, self.rank, self.world, self.device)
        out = self.exchange.step()
        out["scene_familiarity"] = None
        return out

    def clear_library(self):
        if self.exchange is not None:
            self.exchange.close()
            self.exchange = None
        self.engine.clear_library()

    def close(self):
        if self.exchange is not None:
            self.exchange.close()
            self.exchange = None
        self.engine.close()


def device_sharded_sads_familiarity(chem_weight, rank, world_size, device, gather_views=True):
    """Plug-in for navsim_amd.NavBySceneFamiliarity with library AND sensor model on the GPUs, sharded over the ranks
    of an initialised torch.distributed (nccl) group: `NavBySceneFamiliarity(..., familiarity_model=this,
    track_scene_familiarity=False)` on every rank."""
    def model(scenes):
        raise NotImplementedError("this model builds its library on the device from the training path "
                                  "(NavBySceneFamiliarity with use_gpu_sensor=True)")

    def make_engine():
        assert 0 <= chem_weight <= 1
        from .engine import FamiliarityEngine
        import torch
        dev = torch.device(device)
        return ShardedDeviceEngine(FamiliarityEngine(device=dev.index or 0), rank, world_size, dev, gather_views)

    def bind(engine, scenes):
        def func(scene, fambuf):
            raise NotImplementedError("per-view scores of a sharded library live on different ranks")
        func.max_familiarity = scenes[0].shape[0] * scenes[0].shape[1]
        func.engine = engine
        func.chem_weight = chem_weight
        return func

    model.make_engine = make_engine
    model.from_engine = bind
    model.chem_weight = chem_weight
    return model


def step_resident(engine, gather, rank):
    """One sharded step on patches already resident on the device (benchmark / pipelined form)."""
    engine.step_enqueue(want_scene=False)
    res = engine.step_wait(want_scene=False)
    A = len(res["angle_familiarity"])
    records = gather(pack_record(res))
    again, ranks = needs_resolve(records, res["delta"])
    if again:
        if rank in ranks and not (res["flags"] & 3):
            res = engine.resolve()
        records = gather(pack_record(res))
    return merge_records(records, res["delta"], A)
