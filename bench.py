#!/usr/bin/env python3
"""bench.py -- view-comparisons/s of the scene-familiarity hot path on MI355X.

One "step" = one navigation step's scoring: A sensor patches against every stored view of the
library resident in HBM (scoring kernel, per-view/per-heading reductions, tie resolver, result
read-back), i.e. what replaces navsim/NavBySceneFamiliarity.py:283-316 + navsim/util.pyx:31-73.

Workload at N=1: BASELINE.json configs[1] -- 64x64 sensor, 50 000 stored views, 16 headings,
synthetic views (navsim_amd.synth, generated on the device).  With N>1 every rank holds its own
50 000-view shard of an N*50 000-view library (weak scaling) and the per-step exchange is one
all-gather of per-heading records over RCCL (navsim_amd/sharded.py).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
for _p in (REPO, os.path.join(REPO, "navigation-by-deja-vu_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np  # noqa: E402

HBM_PEAK_GBPS = 8000.0      # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s spec, ~6.3 TB/s achievable)


def committed_traffic(workload_key):
    """HBM bytes per launch of the scoring kernel from the committed PMC passes (profiles/*_summary.json).

    bench.py cannot run rocprofv3 on itself; tools/profile_bench.sh collects FETCH_SIZE and WRITE_SIZE in their own
    passes of this same command and tools/summarize_profile.py applies the gfx950 correction (FETCH_SIZE x 2).  The
    number is reported only when the profiled workload is the one being benchmarked.
    """
    import glob
    best = None
    for f in sorted(glob.glob(os.path.join(REPO, "profiles", "r*_summary.json"))):
        try:
            d = json.load(open(f))
            if d["bench_under_trace"]["config"]["workload"] == workload_key:
                best = (float(d["k_sad_tiles"]["hbm_traffic_bytes_per_launch"]), os.path.basename(f))
        except Exception:   # noqa: BLE001
            continue
    return best


def cpu_baseline(h, w, A, cw, seed, budget_views):
    """The oracle (C restatement of util.pyx:31-73, 1 thread) on a bounded sample of the same workload."""
    from navsim_amd import synth
    from oracle import oracle
    lib = synth.synth_views(seed, budget_views, h, w)
    patches = synth.synth_patches(seed, A, h, w)
    fam = np.empty(budget_views)
    oracle.sads_hsv(lib[:64], patches[0], cw, fam[:64].copy())      # warm the library / page in
    t0 = time.perf_counter()
    for a in range(A):
        oracle.sads_hsv(lib, patches[a], cw, fam)
    dt = time.perf_counter() - t0
    model = "unknown"
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    # second figure: the integer form of the same arithmetic on several host cores (OpenMP)
    threads = max(1, min(16, os.cpu_count() or 1))           # the GPU box's CPU share per GPU is 16 threads
    parallel = None
    try:
        oracle.step_fast(lib[:256], patches, cw, threads)       # warm the thread pool
        t0 = time.perf_counter()
        oracle.step_fast(lib, patches, cw, threads)
        dtp = time.perf_counter() - t0
        parallel = dict(value=budget_views * A / dtp, unit="view-comparisons/s", cores=threads, kind="port",
                        sample="same sample, integer sums (oracle_step_fast, -O3 -fopenmp), %.2f s" % dtp)
    except Exception as e:                                       # noqa: BLE001 - the first figure is the contract
        parallel = dict(error=repr(e))
    return dict(value=budget_views * A / dt, unit="view-comparisons/s", cores=1, kind="port", multicore=parallel,
                sample="%d of the stored views x %d headings, %dx%d sensor, chem_weight %g, %.1f s of one host core; "
                       "linear in views (util.pyx:44)" % (budget_views, A, w, h, cw, dt),
                host_cpu=model, host_logical_cores=os.cpu_count())


def ensemble_comparisons_per_s(h, w, A, cw, seed, n_agents, n_views, n_steps):
    """One GPU's share of BASELINE.json configs[4] (256 agents on 8 GPUs, 100k views replicated): n_agents agents x A
    headings per ensemble step through dv_step_batch, patches uploaded every step; one planted answer is checked."""
    import navsim_amd
    from navsim_amd import synth
    eng = navsim_amd.FamiliarityEngine(0)
    try:
        eng.generate_library(seed, n_views, h, w, chem_weight=cw)
        patches = synth.synth_patches(seed, n_agents * A, h, w).reshape(n_agents, A, h, w, 3)
        patches[n_agents // 2, A // 3] = synth.synth_views(seed, 1, h, w, first_view=n_views // 3)[0]
        res = eng.step_batch(patches)
        if (res[n_agents // 2]["best_idex"], res[n_agents // 2]["best_view"]) != (A // 3, n_views // 3):
            raise RuntimeError("planted view not found by the ensemble step")
        t0 = time.perf_counter()
        for _ in range(n_steps):
            eng.step_batch(patches)
        dt = (time.perf_counter() - t0) / n_steps
    finally:
        eng.close()
    return dict(view_comparisons_per_s=n_agents * A * n_views / dt, agent_steps_per_s=n_agents / dt, ms_per_ensemble_step=dt * 1e3,
                what="%d agents x %d headings against %d views (%dx%d), dv_step_batch, patches uploaded each step: one "
                     "GPU's share of BASELINE.json configs[4]" % (n_agents, A, n_views, w, h))


def agent_steps_per_s(h, w, A, cw, n_views, seed, n_steps):
    """Full navsim-style agent on the same shape: sense (GPU) + score + decide + move, per step."""
    import navsim_amd
    from navsim_amd import synth
    L = 2000                                                    # scripts/run_experiment.py:40 mentions 2000x2000 landscapes
    land = synth.synth_landscape(seed, L, 4)
    path = synth.sin_training_path(0.5, 0.2 * L, 0.6 * L, arclen=0.6 * L * 1.4 / n_views)[:n_views]
    nsf = navsim_amd.NavBySceneFamiliarity(land, (w, h), 0.5, n_test_angles=A, n_sensor_levels=5,
                                           familiarity_model=navsim_amd.sads_familiarity(cw),
                                           track_scene_familiarity=False)
    nsf.train_from_path(path)
    d = path[2] - path[1]
    nsf.angle = float(np.arctan2(d[1], d[0]) % (2 * np.pi))
    nsf.position = path[1] + np.array([1.0, -1.0])
    for _ in range(10):
        nsf.step_forward(fake=True)
    t0 = time.perf_counter()
    for _ in range(n_steps):
        nsf.step_forward(fake=True)
    dt = time.perf_counter() - t0
    n_lib = len(path)
    # the same agent as the first of an ensemble of 32 stepping in lockstep (sensing and scoring batched, 64/A agents
    # per library pass; navsim_amd.NavEnsemble): agent-steps per second of the whole ensemble
    ens_rate = None
    try:
        n_ens = 32
        idx = np.linspace(5, n_lib - 50, n_ens).astype(int)
        poses = []
        for i in idx:
            dd = path[i + 1] - path[i]
            poses.append((path[i] + np.array([1.0, -1.0]), float(np.arctan2(dd[1], dd[0]) % (2 * np.pi))))
        ens = navsim_amd.NavEnsemble.from_agent(nsf, poses)
        for _ in range(2):
            ens.step_forward(fake=True)
        t0 = time.perf_counter()
        for _ in range(10):
            ens.step_forward(fake=True)
        ens_rate = n_ens * 10 / (time.perf_counter() - t0)
    except Exception:                                            # noqa: BLE001 - an extra figure only
        ens_rate = None
    nsf.clear_training()
    return n_steps / dt, n_lib, ens_rate


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--views", type=int, default=50000, help="stored views per GPU")
    ap.add_argument("--sensor", type=int, default=64, help="sensor is SENSOR x SENSOR pixels")
    ap.add_argument("--headings", type=int, default=16)
    ap.add_argument("--chem-weight", type=float, default=0.25,
                    help="0 < cw < 1 keeps all three reference bytes per pixel (H,S,V) algorithmically live")
    ap.add_argument("--seed", type=int, default=20261004)
    ap.add_argument("--cpu-views", type=int, default=50000, help="views in the CPU-baseline sample (0 = skip); the "
                    "default is the whole configs[1] library: ~12 s of one host core")
    ap.add_argument("--agent-steps", type=int, default=300, help="steps of the full agent loop timed at N=1 (0 = skip)")
    ap.add_argument("--event-every", type=int, default=4, help="bracket every n-th timed step's scoring kernel with "
                    "HIP events (roofline.kernel_ms); 1 = every step")
    ap.add_argument("--batch-agents", type=int, default=32, help="agents of the ensemble block (configs[4] share of one "
                    "GPU; 0 = skip)")
    ap.add_argument("--exchange", choices=["rccl", "mailbox"], default="rccl", help="per-step exchange between ranks: one RCCL "
                    "all-gather (default), or the single-node mailbox in host-shared memory (no collective kernel)")
    ap.add_argument("--skip-known-answer", action="store_true", help="experiments with deliberately broken kernels only")
    ap.add_argument("--force-dist", action="store_true", help="initialise torch.distributed even with one rank "
                    "(exercises the RCCL exchange path on a single GPU)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N>1 (nccl = RCCL; gloo only to "
                    "rehearse the exchange on a box with fewer GPUs than ranks)")
    args = ap.parse_args()

    # Libraries (RCCL's version banner, HIP warnings) write to fd 1; keep stdout for the ONE JSON line.
    json_fd = os.dup(1)
    os.dup2(2, 1)

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node %d"
                         % (args.gpus, world, args.gpus))

    import torch
    import torch.distributed as dist
    import navsim_amd
    from navsim_amd import sharded

    device_index = local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(device_index)
    use_dist = world > 1 or args.force_dist
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", device_index))
            gather = sharded.torch_gather(device=torch.device("cuda", device_index))
        else:
            dist.init_process_group(backend=args.backend)
            gather = sharded.torch_gather(device=None)
    else:
        gather = None

    F, h, w, A, cw = args.views, args.sensor, args.sensor, args.headings, args.chem_weight
    eng = navsim_amd.FamiliarityEngine(device=device_index)
    eng.generate_library(args.seed, F, h, w, cw, first_view=rank * F)     # this rank's shard, made in HBM
    eng.generate_patches(args.seed, A)                                     # same patches on every rank
    info = eng.library_info()

    exchange = None
    if use_dist and args.exchange == "mailbox":
        exchange = sharded.MailboxExchange(eng, rank, world)
    elif use_dist and args.backend == "nccl":
        exchange = sharded.DeviceExchange(eng, rank, world, torch.device("cuda", device_index))

    def one_step():
        if exchange is not None:
            return exchange.step()
        if use_dist:
            return sharded.step_resident(eng, gather, rank)
        eng.step_enqueue(want_scene=False)
        return eng.step_wait(want_scene=False)

    def fence():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()
        eng.synchronize()

    for _ in range(args.warmup):
        one_step()
    # HIP events around the scoring kernel on a sample of the timed steps: a pair costs ~5 us of stream time
    eng.profile_kernel(True, every=args.event_every)
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res = one_step()
    fence()
    dt = time.perf_counter() - t0
    kern_ms_total, kern_n = eng.profile_read()
    eng.profile_kernel(False)

    # Known-answer step, outside the timed region: heading a* looks at a view stored on the LAST rank, so the global
    # decision has to come through the exchange (a stale or mis-ordered record gives the previous answer instead).
    from navsim_amd import synth
    a_star, f_star = A // 2, (world - 1) * F + (12345 % F)
    probe = synth.synth_patches(args.seed, A, h, w)
    probe[a_star] = synth.synth_views(args.seed, 1, h, w, first_view=f_star)[0]
    eng.upload_patches(probe)
    chk = one_step()
    ok = args.skip_known_answer or ((int(chk["best_idex"]), int(chk["best_view"])) == (a_star, f_star) and
                                    float(chk["step_familiarity"]) == float(h * w))
    if not ok:
        sys.stderr.write("rank %d: known-answer step FAILED\n" % rank)
    if use_dist:                                                 # fail together: nobody is left waiting in a collective
        flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device="cuda" if args.backend == "nccl" else "cpu")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        all_ok = bool(flag.item())
    else:
        all_ok = ok
    if not all_ok:
        raise SystemExit("rank %d: known-answer step failed: got heading %d view %d familiarity %r, expected %d %d %r"
                         % (rank, chk["best_idex"], chk["best_view"], chk["step_familiarity"], a_star, f_star, float(h * w)))

    if use_dist:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    if rank == 0:
        comparisons = float(world) * F * A * args.steps
        kern_ms = kern_ms_total / max(kern_n, 1)
        # Algorithmic bytes per launch, SURVEY.md section 8(d): F*P*s with s = 3 for sads_hsv with chem_weight > 0
        # (H,S,V all live) and s = 1 for chem_weight = 0 (V only); one launch reads the library once for all headings.
        s_ref = 3 if cw > 0 else 1
        algo_bytes = float(F) * h * w * s_ref
        achieved = algo_bytes / (kern_ms * 1e-3) / 1e9
        # What this build actually keeps in HBM and streams per launch: when the library has two
        # hues and no saturation above 127, one signed plane carries (H,S), i.e. 2 bytes per pixel instead of 3.
        stored_bytes = float(F) * h * w * info["n_planes"]
        achieved_stored = stored_bytes / (kern_ms * 1e-3) / 1e9
        named = {(64, 50000, 16): " (BASELINE.json configs[1])", (128, 500000, 32): " (BASELINE.json configs[2])"}
        workload = ("%dx%d sensor, %d stored views per GPU, %d headings, sads_hsv chem_weight=%g%s"
                    % (w, h, F, A, cw, named.get((w, F, A), "") if w == h else ""))
        traffic = committed_traffic(workload)
        # SURVEY.md 8(d) asks for both peaks: the spec figure and what a pure streaming read reaches on this device
        try:
            read_ceiling = float(eng.stream_read_gbps(1 << 30, 10))
        except Exception:                                        # measurement aid only
            read_ceiling = None
        out = {
            "metric": "view-comparisons/sec (sensor x library x headings)",
            "value": comparisons / dt,
            "unit": "view-comparisons/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u8",
            "data": "synthetic",
            "config": {
                "workload": workload,
                "views_per_gpu": F, "total_views": world * F, "headings": A, "sensor": [w, h],
                "workgroup_shape": eng.workgroup_shape(A),
                "bytes_per_pixel_reference": s_ref, "bytes_per_pixel_stored": info["n_planes"], "parallelism": "library sharded x%d" % world,
                "exchange": "none" if not use_dist else "1 all-gather of per-heading records per step (%s)" % (
                    "mailbox in host-shared memory, no collective" if args.exchange == "mailbox" else
                    (("RCCL ncclAllGather on the step's stream, device-resident"
                      if (exchange is not None and exchange.direct is not None) else "RCCL via torch.distributed, device-resident")
                     if args.backend == "nccl" else args.backend)),
            },
            "nav_steps_per_s": args.steps / dt,
            "best_heading": int(res["best_idex"]),
            "known_answer_step": "SKIPPED" if args.skip_known_answer else "ok on every rank (heading %d, view %d of %d, through the exchange)" % (a_star, f_star, world * F),
            "roofline": {
                "bound": "hbm", "kernel": "k_sad_tiles", "achieved": achieved, "peak": HBM_PEAK_GBPS,
                "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
                "traffic": traffic[0] if traffic else None,
                "traffic_source": ("PMC FETCH_SIZE x2 + WRITE_SIZE per launch, " + traffic[1]) if traffic else None,
                "algorithmic_bytes_per_launch": algo_bytes, "kernel_ms": kern_ms, "launches_timed": kern_n,
                "stored_bytes_per_launch": stored_bytes, "achieved_stored": achieved_stored,
                "frac_stored": achieved_stored / HBM_PEAK_GBPS,
                "measured_read_ceiling": read_ceiling,
                "frac_stored_of_measured_ceiling": (achieved_stored / read_ceiling) if read_ceiling else None,
            },
        }
        if world == 1 and args.agent_steps > 0 and not args.force_dist:
            eng.clear_library()                                  # make room: the agent builds its own library
            sps, n_lib, ens_rate = agent_steps_per_s(h, w, A, cw, F, args.seed, args.agent_steps)
            out["agent"] = {"nav_steps_per_s": sps, "view_comparisons_per_s": sps * n_lib * A, "library_views": n_lib,
                            "ensemble_of_32_nav_steps_per_s": ens_rate,
                            "what": "navsim_amd.NavBySceneFamiliarity.step_forward(fake=True): sensor model on the GPU "
                                    "(2000x2000 landscape resident), scoring, decision, position update; Python caller"}
        if world == 1 and args.batch_agents > 0 and not args.force_dist:
            try:
                out["ensemble"] = ensemble_comparisons_per_s(h, w, A, cw, args.seed, args.batch_agents, 100000, 5)
            except Exception as e:                               # noqa: BLE001 - an extra block must not cost the JSON line
                out["ensemble"] = {"error": repr(e)}
        if world == 1 and args.cpu_views > 0:
            out["cpu_baseline"] = cpu_baseline(h, w, A, cw, args.seed, min(args.cpu_views, F))
        else:
            out["cpu_baseline"] = None
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if exchange is not None:
        exchange.close()
    eng.close()
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
