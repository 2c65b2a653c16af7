#!/usr/bin/env python3
"""bench.py -- view-comparisons/s of the scene-familiarity hot path on MI355X.

One "step" = one navigation step's scoring: A sensor patches against every stored view of the
library resident in HBM (patch preparation, scoring kernel, per-view/per-heading reductions, tie
resolver, result read-back), i.e. what replaces navsim/NavBySceneFamiliarity.py:283-316 +
navsim/util.pyx:31-73.  Every timed step scores FRESH patches (generated on the device from seed + step, as the
reference senses new patches at every step, NavBySceneFamiliarity.py:289-299), so the per-step preparation
kernel (k_patch_prep: patches, coefficient images, constants) is inside the timed region; `scoring_only` repeats the measurement on
resident patches (the figure rounds 1-2 reported as the headline).

Workload at N=1: BASELINE.json configs[2] -- 128x128 sensor, 500 000 stored views, 32 headings (the largest
single-GPU configuration), synthetic views (navsim_amd.synth, generated on the device).  With N>1 every rank holds
its own 500 000-view shard of an N*500 000-view library -- at N=8 that is configs[3], 4 M views -- and the per-step
exchange is one RCCL collective (navsim_amd/sharded.py).  configs[1] (64x64, 50 000 views, 16 headings), the agent
loop, the ensemble share of configs[4] and the ssd_f32 metric are timed after the headline as secondary blocks.

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
FP4_MFMA_PEAK_TOPS = 10000.0  # dense fp4 (f8f6f4) peak, twice the int8 figure (MI355X_MICROARCH.md, Matrix cores)
I8_MFMA_PEAK_TOPS = 5000.0  # dense int8 matrix-core peak: 2x the ~2.5 PF dense bf16 figure (MI355X_MICROARCH.md, Matrix cores)
CLOCK_PEAK_GHZ = 2.4


def kernel_of_shape(shape, generic=False):
    if generic:
        return "k_sad_generic"
    return {5: "k_sad_packed", 6: "k_sad_mfma_dual"}.get(shape, "k_sad_tiles")


def launch_ranks(n):
    """`python bench.py --gpus N` without a launcher: start N copies of this script, one per GPU, with the environment
    torch.distributed.run would give them (before this process makes any GPU call), and pass rank 0's line through."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    out, _ = procs[0].communicate()
    codes = [procs[0].returncode] + [q.wait() for q in procs[1:]]
    sys.stdout.buffer.write(out)
    sys.stdout.flush()
    raise SystemExit(max(abs(c) for c in codes))


def committed_traffic(workload_key, kernel="k_sad_tiles"):
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
                k = kernel if kernel in d else {"k_sad_mfma_dual": "k_sad_mfma"}.get(kernel, kernel)   # (summaries key the dual kernel as k_sad_mfma)
                best = (float(d[k]["hbm_traffic_bytes_per_launch"]), os.path.basename(f))
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
    headings per ensemble step against n_views views.  Two forms: `sensed` -- the agents' patches are sensed on the device
    from the resident landscape (dv_sense_step_batch: nothing but poses goes up), the form an ensemble run uses -- and
    `uploaded` -- patches handed over as host buffers every step (dv_step_batch, PCIe-inclusive); one planted answer is
    checked there."""
    import navsim_amd
    from navsim_amd import synth
    out = {}
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
        out["uploaded"] = dict(view_comparisons_per_s=n_agents * A * n_views / dt, ms_per_ensemble_step=dt * 1e3,
                               what="dv_step_batch, %d bytes of patches uploaded each step" % patches.nbytes)
    finally:
        eng.close()
    # sensed: the agents' patches come from the sensor model on the device (landscape resident, poses spread along a path); the
    # library is the same synthetic one (views unrelated to the landscape: the work per step is what it is for any library of
    # these levels, without the exact ties a library sensed every 0.02 px along one path would add)
    L = 2000
    land = synth.synth_landscape(seed, L, 4)
    path = synth.sin_training_path(0.5, 0.2 * L, 0.6 * L, arclen=8.0)
    nsf = navsim_amd.NavBySceneFamiliarity(land, (w, h), 0.5, n_test_angles=A, n_sensor_levels=5,
                                           familiarity_model=navsim_amd.sads_familiarity(cw), track_scene_familiarity=False)
    e2 = nsf._engine
    try:
        e2.generate_library(seed, n_views, h, w, chem_weight=cw)
        idx = np.linspace(5, len(path) - 5, n_agents).astype(int)
        xs, ys, angs = [], [], []
        for i in idx:
            dd = path[i + 1] - path[i]
            nsf.position = tuple(path[i] + np.array([1.0, -1.0]))
            nsf.angle = float(np.arctan2(dd[1], dd[0]) % (2 * np.pi))
            x, y, a = nsf.headings_to_test()
            xs.append(x); ys.append(y); angs.append(a)
        angs = np.stack(angs)
        info = e2.library_info()
        planes = (info["bit_planes_hs"] + info["bit_planes_v"]) if info["has_bit_planes"] else 0
        for _ in range(3):
            e2.sense_step_batch(xs, ys, angs)
        resolved = sum(1 for r in e2.sense_step_batch(xs, ys, angs) if r["flags"] & 1)
        form = e2.scoring_form()
        t0 = time.perf_counter()
        for _ in range(2 * n_steps):
            e2.sense_step_batch(xs, ys, angs)
        dt = (time.perf_counter() - t0) / (2 * n_steps)
        out["sensed"] = dict(view_comparisons_per_s=n_agents * A * n_views / dt, ms_per_ensemble_step=dt * 1e3, library_views=n_views,
                             agents_resolved_exactly=resolved, kernel_form=form,
                             what="dv_sense_step_batch: patches sensed on the device, poses only go up")
    finally:
        e2.close()
    best = out["sensed"]
    k_elems = float(planes) * h * w                                # K-elements per (view, heading): bit planes x pixels of the sensed library
    # the sensed patches are 5-level sensor output scored against a synthetic library: on its levels they take the fp4 form, off them
    # the int8 form at half the peak -- the fraction is priced against the form that ran
    fp4 = bool(out["sensed"]["kernel_form"].get("fp4"))
    peak = FP4_MFMA_PEAK_TOPS if fp4 else I8_MFMA_PEAK_TOPS
    return dict(view_comparisons_per_s=best["view_comparisons_per_s"], agent_steps_per_s=n_agents / (best["ms_per_ensemble_step"] * 1e-3),
                ms_per_ensemble_step=best["ms_per_ensemble_step"], sensed=out["sensed"], uploaded=out["uploaded"],
                kernel="k_sad_lc22 (two view groups x two heading tiles per consumer; DEJAVU_LC22=0: k_sad_mfma_dual)",
                mfma_form="fp4" if fp4 else "int8", mfma_peak_tops=peak,
                mfma_frac_of_peak=2.0 * best["view_comparisons_per_s"] * k_elems / 1e12 / peak,
                what="%d agents x %d headings against %d views (%dx%d): one GPU's share of BASELINE.json configs[4]; headline = "
                     "the sensed form; mfma_frac = 2 x comparisons x bit planes (%d) x pixels / s over the dense peak of the form that ran "
                     "(fp4 10 POP/s, int8 5)" % (n_agents, A, n_views, w, h, planes))


def agent_steps_per_s(h, w, A, cw, n_views, seed, n_steps):
    """Full navsim-style agent on the configs[1] shape: sense (GPU) + score + decide + move + error metrics, per step:
    step_forward() exactly as scripts/run_experiment.py:243-245 calls it (fake=False, nothing skipped)."""
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
    rates = {}
    outliers = {}
    # The interpreter's cyclic collector stays ON during the timed loops; what is taken out of its reach is the object graph that
    # exists before them (gc.freeze: this process imports torch, ~10^6 container objects that a full collection walks in ~45 ms --
    # one such pause inside a 3000-step loop is 25 % of it and says nothing about the step)
    import gc
    gc.collect()
    gc.freeze()
    for fake in (True, False):
        nsf.position = path[1] + np.array([1.0, -1.0])
        nsf.angle = float(np.arctan2(d[1], d[0]) % (2 * np.pi))
        nsf.reset_error()
        done = 0
        durs = []
        try:
            for _ in range(100):                                # (clocks and caches settled: 300 timed steps gave 13 500 where 3 000 give 14 100)
                nsf.step_forward(fake=fake)
            t0 = time.perf_counter()
            for _ in range(n_steps):
                t1 = time.perf_counter()
                nsf.step_forward(fake=fake)
                durs.append(time.perf_counter() - t1)
                done += 1
            rates[fake] = done / (time.perf_counter() - t0)
            outliers[fake] = dict(median_step_us=float(np.median(durs)) * 1e6,
                                  steps=[(i, int(x * 1e6)) for i, x in enumerate(durs) if x > 4 * float(np.median(durs))][:8])
        except navsim_amd.StopNavigationException:              # left the path before n_steps: rate over what ran
            rates[fake] = done / max(time.perf_counter() - t0, 1e-9) if done else None
    gc.unfreeze()
    n_lib = len(path)
    # the same agent as the first of an ensemble of 32 stepping in lockstep (sensing and scoring batched, 64/A agents
    # per library pass; navsim_amd.NavEnsemble): agent-steps per second of the whole ensemble
    ens_rate = ens_rate_real = None
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
        # ... and with the error metrics on (fake=False: every member's update_error, one device call per ensemble step)
        for _ in range(2):
            ens.step_forward()
        t0 = time.perf_counter()
        for _ in range(10):
            ens.step_forward()
        ens_rate_real = n_ens * 10 / (time.perf_counter() - t0)
    except Exception:                                            # noqa: BLE001 - extra figures only
        pass
    nsf.clear_training()
    return rates, n_lib, (ens_rate, ens_rate_real), outliers


def ssd_f32_block(device_index, F, h, w, A, steps, seed=4242):
    """The north star's literal "fp32 SSD": float32 single-channel views, the metric the reference defines as `ssds`
    (util.pyx:171-184) -- on configs[1]'s shape (819 MB) and on configs[2] as BASELINE.json words it (500 000 x 128x128 float32 =
    32.8 GB, 32 headings).  The library is generated on the device (dv_generate_library_f32); patches are uploaded every step
    (PCIe-inclusive step).  Steps without per-view output take the cross-term form on the fp32 matrix cores (k_ssd_f32_mfma), the
    listed candidates re-scored exactly; `direct` repeats the timing with DEJAVU_SSD_MFMA=0 (k_ssd_tiles, 16 headings per pass)."""
    import navsim_amd
    from navsim_amd import synth
    rng = np.random.default_rng(1)
    patches = rng.random((A, h, w), dtype=np.float32)
    plant = 31337 % F
    patches[A // 2] = synth.synth_views_f32(seed, 1, h, w, first_view=plant)[0] + np.float32(0.01)
    out = {}
    for name, env in (("mfma", None), ("direct", "0")):
        if env is not None:
            os.environ["DEJAVU_SSD_MFMA"] = env
        try:
            eng = navsim_amd.FamiliarityEngine(device_index)
        finally:
            os.environ.pop("DEJAVU_SSD_MFMA", None)
        try:
            eng.generate_library_f32(seed, F, h, w)
            n = steps if name == "mfma" else max(steps // 4, 5)
            for _ in range(3):
                r = eng.step_f32(patches)
            if (r["best_idex"], r["best_view"]) != (A // 2, plant):
                raise RuntimeError("ssd_f32: planted view not found")
            eng.profile_kernel(True)
            t0 = time.perf_counter()
            for _ in range(n):
                eng.step_f32(patches)
            dt = (time.perf_counter() - t0) / n
            kms, kn = eng.profile_read()
            eng.profile_kernel(False)
            out[name] = (dt, kms / max(kn, 1), kn, int(r["n_candidates"]))
        finally:
            eng.close()
    dt, kern_ms, kn, ncand = out["mfma"]
    passes = (A + 31) // 32 if A > 16 else 1
    algo = float(F) * h * w * 4
    streamed = algo * passes
    return {"workload": "%dx%d sensor, %d stored float32 views, %d headings, ssd_f32" % (w, h, F, A), "dtype": "f32",
            "value": F * A / dt, "unit": "view-comparisons/s", "ms_per_step": dt * 1e3, "steps": steps,
            "candidates_rescored_exactly": ncand,
            "direct_form": {"ms_per_step": out["direct"][0] * 1e3, "kernel_ms": out["direct"][1], "kernel": "k_ssd_tiles",
                            "library_passes": (A + 15) // 16},
            "roofline": {"bound": "hbm", "kernel": "k_ssd_f32_mfma<16>" if A <= 16 else "k_ssd_f32_bf16x2", "kernel_ms": kern_ms, "launches_timed": kn,
                         "library_passes": passes, "achieved": streamed / (kern_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBPS,
                         "unit": "GB/s", "frac": streamed / (kern_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                         "bytes_basis": "streamed library bytes (4 B/px x passes of 32 headings) = the algorithmic bytes of SURVEY 8(d), s = 4",
                         "algorithmic_bytes_per_launch": algo,
                         "frac_algorithmic": algo / (kern_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                         "mfma": ({"dtype": "f32 (v_mfma_f32_16x16x1_4b_f32)", "ops_per_launch": 2.0 * F * h * w * 16, "peak": 157.3, "unit": "TFLOP/s",
                                   "frac": 2.0 * F * h * w * 16 / (kern_ms * 1e-3) / 1e12 / 157.3} if A <= 16 else
                                  {"dtype": "bf16 x 3 terms (v_mfma_f32_32x32x16_bf16: lh ph + lh pl + ll ph)", "ops_per_launch": 6.0 * F * h * w * 32 * passes,
                                   "peak": 2500.0, "unit": "TFLOP/s", "frac": 6.0 * F * h * w * 32 * passes / (kern_ms * 1e-3) / 1e12 / 2500.0})}}


def ssd_u8_block(device_index, F, h, w, A, steps):
    """The exact-integer SSD of uint8 views on the int8 matrix cores (`ssds`, util.pyx:171-184, on uint8 data):
    one byte per pixel per pass of 32 headings.  Patches are uploaded every step."""
    import navsim_amd
    rng = np.random.default_rng(1)
    lib = rng.integers(0, 256, (F, h, w), dtype=np.uint8)
    patches = rng.integers(0, 256, (A, h, w), dtype=np.uint8)
    patches[A // 2] = lib[31337 % F]
    patches[A // 2, 0, 0] ^= 1
    eng = navsim_amd.FamiliarityEngine(device_index)
    try:
        eng.set_library_u8(lib)
        for _ in range(5):
            r = eng.step_u8(patches)
        if (r["best_idex"], r["best_view"], r["step_ssd"]) != (A // 2, 31337 % F, 1.0):
            raise RuntimeError("ssd_u8: planted view not found")
        eng.profile_kernel(True)
        t0 = time.perf_counter()
        for _ in range(steps):
            eng.step_u8(patches)
        dt = (time.perf_counter() - t0) / steps
        kms, kn = eng.profile_read()
        eng.profile_kernel(False)
    finally:
        eng.close()
    passes = (A + 31) // 32
    kern_ms = kms / max(kn, 1)
    streamed = float(F) * h * w * passes
    return {"workload": "%dx%d sensor, %d stored uint8 views, %d headings, ssd_u8 (exact, int8 matrix cores)" % (w, h, F, A), "dtype": "u8",
            "value": F * A / dt, "unit": "view-comparisons/s", "ms_per_step": dt * 1e3, "steps": steps,
            "roofline": {"bound": "hbm", "kernel": "k_ssd_u8_mfma", "kernel_ms": kern_ms, "launches_timed": kn,
                         "library_passes": passes, "achieved": streamed / (kern_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBPS,
                         "unit": "GB/s", "frac": streamed / (kern_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                         "bytes_basis": "streamed library bytes (1 B/px x passes of 32 headings) = the algorithmic bytes of a uint8 library"}}


def workload_name(w, h, F, A, cw, world):
    named = ""
    if w == h:
        if (w, F, A) == (64, 50000, 16):
            named = " (BASELINE.json configs[1])"
        elif (w, F, A) == (128, 500000, 32):
            named = " (BASELINE.json configs[2])" if world == 1 else (
                " (BASELINE.json configs[3]: %d views over %d GPUs)" % (world * F, world) if world == 8 else
                " (per-rank share of BASELINE.json configs[3], %d ranks)" % world)
    return "%dx%d sensor, %d stored views per GPU, %d headings, sads_hsv chem_weight=%g%s" % (w, h, F, A, cw, named)


def roofline_block(eng, info, shape, kern_ms, kern_n, F, h, w, A, cw, workload, with_ceiling=True):
    """Roofline of the scoring kernel.  `achieved` / `frac` are on the bytes the kernel MOVES: the HBM traffic of the
    committed PMC passes when they are of this workload, else the bytes it streams by construction (stored library
    bytes + the partial sums it writes).  The SURVEY section 8(d) figure (F*P*s reference bytes) is kept as
    `algorithmic_*`: it counts 3 B/px where the bit-plane layout stores 0.75."""
    kernel = kernel_of_shape(shape, bool(info.get("generic_hue")))
    s_ref = 3 if cw > 0 else 1
    algo_bytes = float(F) * h * w * s_ref
    streamed = float(info["bit_tile_bytes"] if shape == 6 else info["tile_bytes"])
    if shape == 6 and info.get("mixed_layout"):                  # + the saturation byte planes' share of the byte tiles
        streamed += float(info["tile_bytes"]) * info["n_hue_planes"] / max(info["n_planes"], 1)
    nsum = (1 if info["n_hue_planes"] > 0 or info["generic_hue"] else 0) + (1 if info["has_value_plane"] else 0)
    apad = 8 if A <= 8 else (16 if A <= 16 else (32 if A <= 32 else 64))
    traffic = committed_traffic(workload, kernel)
    if traffic and shape == 6 and info.get("mixed_layout"):      # two scoring kernels per step: + the saturation byte pass
        t2 = committed_traffic(workload, "k_sad_tiles")
        traffic = (traffic[0] + t2[0], traffic[1] + " (k_sad_mfma_dual + k_sad_tiles)") if t2 else None
    moved = traffic[0] if traffic else None
    t = kern_ms * 1e-3
    # partial sums written once per chunk; their count is the engine's choice, so only a lower bound (one chunk) is
    # known here -- the PMC figure, when present, has the real number
    try:
        form = eng.scoring_form()
    except Exception:                                            # an engine without the call
        form = dict(matrix_cores=shape == 6, fp4=False, fused_finish=False)
    if shape == 6 and form["fp4"] and info.get("code_tile_bytes"):
        streamed = float(info["code_tile_bytes"])                # the fp4 form reads the value plane as 3-bit level codes
    constructed = streamed + (0.0 if form["fused_finish"] else float(nsum) * apad * F * 4)
    basis = moved if moved is not None else constructed
    out = {
        "bound": "hbm", "kernel": kernel, "achieved": basis / t / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
        "frac": basis / t / 1e9 / HBM_PEAK_GBPS,
        "traffic": moved, "traffic_source": ("PMC FETCH_SIZE x2 + WRITE_SIZE per launch, " + traffic[1]) if traffic else None,
        "bytes_basis": "pmc traffic" if moved is not None else "streamed library bytes + partial sums (by construction)",
        "kernel_form": form,
        "streamed_library_bytes_per_launch": streamed, "kernel_ms": kern_ms, "launches_timed": kern_n,
        "algorithmic_bytes_per_launch": algo_bytes, "algorithmic_achieved": algo_bytes / t / 1e9,
        "frac_algorithmic": algo_bytes / t / 1e9 / HBM_PEAK_GBPS,
    }
    if shape == 6:
        # int8 MFMA work actually issued: 32 heading rows x 32 views x 32 K per instruction, K = planes x pixels
        # rounded up to 256 per segment; one pass per 32 resident headings
        k_total = float(info["bit_tile_bytes"]) / (((F + 63) // 64) * 64 / 32.0) / 1024.0 * 256.0
        passes = (apad + 31) // 32
        ops = 2.0 * 32 * (((F + 63) // 64) * 64) * k_total * passes
        # the fp4 form (on-level patches) multiplies the same K-elements with v_mfma_f32_32x32x64_f8f6f4: twice the int8 peak
        peak = FP4_MFMA_PEAK_TOPS if form["fp4"] else I8_MFMA_PEAK_TOPS
        if info.get("mixed_layout"):
            lane_ops = float(apad) * F * h * w * info["n_hue_planes"] / 4.0
            vpeak = 256 * 4 * 64 / 4.0 * CLOCK_PEAK_GHZ * 1e9
            out["valu"] = {"lane_ops_per_launch": lane_ops, "achieved": lane_ops / t, "peak": vpeak, "unit": "v_sad_u8 lane-ops/s",
                           "frac": lane_ops / t / vpeak, "clock_ghz_assumed": CLOCK_PEAK_GHZ,
                           "what": "the saturation byte planes' pass (k_sad_tiles) behind the matrix-core pass on the value bits: the bound of this layout"}
        out["mfma"] = {"dtype": "fp4 (E2M1 signs x library bits, f32 accumulate, exact)" if form["fp4"] else "i8",
                       "ops_per_launch": ops, "achieved": ops / t / 1e12, "peak": peak,
                       "unit": "TOP/s", "frac": ops / t / 1e12 / peak,
                       "useful_ops_per_launch": 2.0 * A * F * h * w * (info["bit_planes_hs"] + info["bit_planes_v"])}
    else:
        # v_sad_u8 issues one wave64 instruction per SIMD every 4 cycles: 256 CUs x 4 SIMDs x 64 lanes / 4
        lane_ops = float(apad) * F * h * w * info["n_planes"] / 4.0
        peak = 256 * 4 * 64 / 4.0 * CLOCK_PEAK_GHZ * 1e9
        out["valu"] = {"lane_ops_per_launch": lane_ops, "achieved": lane_ops / t, "peak": peak, "unit": "v_sad_u8 lane-ops/s",
                       "frac": lane_ops / t / peak, "clock_ghz_assumed": CLOCK_PEAK_GHZ}
        if info["generic_hue"]:
            # k_sad_generic selects |S_s - S_f| or S_s + S_f per byte by comparing the hue bytes: per library dword and heading the
            # built kernel issues 13 full-rate vector instructions for the H, S pair (xor, 6 v_bitop3, shift, subtract, add, 3 v_sad_u8:
            # llvm-objdump of the shipped code object) and one v_sad_u8 for V -- its instruction stream, not the byte count, is the bound
            per_dword = 13.0 * (1 if info["n_planes"] >= 2 else 0) + (1.0 if info["has_value_plane"] else 0.0)
            lane_ops = float(apad) * F * h * w * per_dword / 4.0
            out["valu"] = {"lane_ops_per_launch": lane_ops, "achieved": lane_ops / t, "peak": peak, "unit": "full-rate VALU lane-ops/s",
                           "frac": lane_ops / t / peak, "clock_ghz_assumed": CLOCK_PEAK_GHZ, "instructions_per_dword_and_heading": per_dword,
                           "what": "k_sad_generic: hue compare + select + three v_sad_u8 per H, S dword and heading, one v_sad_u8 per V dword"}
    if with_ceiling:
        try:
            out["measured_read_ceiling"] = float(eng.stream_read_gbps(1 << 30, 10))
            # the same bytes against what an LDS-DMA stream probe inside this library reads on THIS box (not the spec peak)
            out["frac_of_measured_ceiling"] = out["achieved"] / out["measured_read_ceiling"]
        except Exception:                                        # measurement aid only
            out["measured_read_ceiling"] = None
            out["frac_of_measured_ceiling"] = None
    return out


def generic_block(device_index, seed, F, h, w, A, cw, steps):
    """A library the one-hot layout cannot hold -- more than four hues, full-range S and V (uploaded: random bytes) -- scored by
    k_sad_generic on H, S, V byte planes; resident patches, scoring + reductions per step."""
    import navsim_amd
    from navsim_amd import synth
    lib = synth.random_hsv(seed, (F, h, w, 3))
    patches = synth.random_hsv(seed + 1, (A, h, w, 3))
    patches[A // 2] = lib[F // 3]
    eng = navsim_amd.FamiliarityEngine(device=device_index)
    try:
        eng.set_library(lib, cw)
        del lib
        info = eng.library_info()
        eng.upload_patches(patches)
        for _ in range(5):
            eng.step_enqueue(want_scene=False)
            r = eng.step_wait(want_scene=False)
        if (r["best_idex"], r["best_view"]) != (A // 2, F // 3):
            raise RuntimeError("generic block: planted view not found")
        eng.profile_kernel(True, every=2)
        eng.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            eng.step_enqueue(want_scene=False)
            eng.step_wait(want_scene=False)
        eng.synchronize()
        dt = time.perf_counter() - t0
        kms, kn = eng.profile_read()
        eng.profile_kernel(False)
        shape = eng.workgroup_shape(A)
        workload = "%dx%d sensor, %d stored views, %d headings, sads_hsv chem_weight=%g, uniformly random H, S, V bytes (generic hue planes)" % (w, h, F, A, cw)
        return {"workload": workload, "value": F * A * steps / dt, "unit": "view-comparisons/s", "ms_per_step": dt / steps * 1e3, "steps": steps,
                "generic_hue": bool(info["generic_hue"]), "patches": "resident (uploaded once)",
                "roofline": roofline_block(eng, info, shape, kms / max(kn, 1), kn, F, h, w, A, cw, workload, with_ceiling=False)}
    finally:
        eng.close()


def secondary_scoring(device_index, seed, F, h, w, A, cw, steps, warmup, full_range_s=False, env=None):
    """One more workload on a fresh engine (N=1 only): value, step and kernel time, roofline.  full_range_s: the library's
    saturation takes every value 0..127 (a swept concentration range) -- too many levels for thermometer planes, so the layout
    is the mixed one: value bit planes on the matrix cores, then the saturation byte planes with v_sad_u8, k_finish on both."""
    import navsim_amd
    before = {k: os.environ.get(k) for k in (env or {})}
    os.environ.update(env or {})
    try:
        eng = navsim_amd.FamiliarityEngine(device=device_index)     # (the context reads its knobs when it is created)
    finally:
        for k, v in before.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    try:
        eng.generate_library(seed, F, h, w, cw, full_range_s=full_range_s)
        eng.generate_patches(seed, A)
        info = eng.library_info()
        for i in range(warmup):
            eng.generate_patches(seed + 1 + i, A)
            eng.step_enqueue(want_scene=False)
            eng.step_wait(want_scene=False)
        eng.profile_kernel(True, every=4)
        eng.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            eng.generate_patches(seed + 1000 + i, A)                 # fresh patches: the preparation kernels run every step
            eng.step_enqueue(want_scene=False)
            eng.step_wait(want_scene=False)
        eng.synchronize()
        dt = time.perf_counter() - t0
        kms, kn = eng.profile_read()
        eng.profile_kernel(False)
        t0 = time.perf_counter()
        for _ in range(steps):                                       # the same patches again and again: scoring alone
            eng.step_enqueue(want_scene=False)
            eng.step_wait(want_scene=False)
        eng.synchronize()
        dt_res = time.perf_counter() - t0
        shape = eng.workgroup_shape(A)
        workload = workload_name(w, h, F, A, cw, 1) + (", saturation 0..127 (mixed layout)" if full_range_s else "")
        return {"workload": workload, "value": F * A * steps / dt, "unit": "view-comparisons/s", "ms_per_step": dt / steps * 1e3,
                "steps": steps, "workgroup_shape": shape, "env": env,
                "scoring_only": {"value": F * A * steps / dt_res, "ms_per_step": dt_res / steps * 1e3,
                                 "what": "the same resident patches every step: no preparation kernels"},
                "roofline": roofline_block(eng, info, shape, kms / max(kn, 1), kn, F, h, w, A, cw, workload, with_ceiling=False)}
    finally:
        eng.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--views", type=int, default=500000, help="stored views per GPU")
    ap.add_argument("--sensor", type=int, default=128, help="sensor is SENSOR x SENSOR pixels")
    ap.add_argument("--headings", type=int, default=32)
    ap.add_argument("--chem-weight", type=float, default=0.25,
                    help="0 < cw < 1 keeps all three reference bytes per pixel (H,S,V) algorithmically live")
    ap.add_argument("--full-range-s", action="store_true", help="headline library with saturation 0..127 instead of {0, 127}: the "
                    "mixed layout (value bit planes + saturation bytes); the default run times it as the `full_range_s` block")
    ap.add_argument("--seed", type=int, default=20261004)
    ap.add_argument("--cpu-views", type=int, default=8000, help="views in the CPU-baseline sample (0 = skip); the "
                    "default is ~15 s of one host core at the headline shape")
    ap.add_argument("--agent-steps", type=int, default=2000, help="steps of the full agent loop timed at N=1 (0 = skip)")
    ap.add_argument("--event-every", type=int, default=1, help="bracket every n-th timed step's scoring kernel with "
                    "HIP events (roofline.kernel_ms); an event pair costs ~5 us of stream time")
    ap.add_argument("--batch-agents", type=int, default=32, help="agents of the ensemble block (configs[4] share of one "
                    "GPU; 0 = skip)")
    ap.add_argument("--secondary", type=int, default=1, help="1: also time configs[1] (64x64, 50 000 views, 16 headings) "
                    "at N=1; 0 = skip")
    ap.add_argument("--exchange", choices=["rccl", "mailbox"], default="rccl", help="per-step exchange between ranks: one RCCL "
                    "collective (default), or the single-node mailbox in host-shared memory (no collective kernel)")
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
        if "WORLD_SIZE" not in os.environ and args.gpus > 1:
            os.dup2(json_fd, 1)
            launch_ranks(args.gpus)                                  # does not return
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
    eng.generate_library(args.seed, F, h, w, cw, first_view=rank * F, full_range_s=args.full_range_s)     # this rank's shard, made in HBM
    eng.generate_patches(args.seed, A)                                     # same patches on every rank
    info = eng.library_info()

    exchange = None
    if use_dist and args.exchange == "mailbox":
        exchange = sharded.MailboxExchange(eng, rank, world)
    elif use_dist and args.backend == "nccl":
        exchange = sharded.DeviceExchange(eng, rank, world, torch.device("cuda", device_index))

    def one_step(fresh=None):
        if fresh is not None:
            eng.generate_patches(fresh, A)            # new patches in HBM (same on every rank): k_patch_prep
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

    for i in range(args.warmup):
        one_step(args.seed + 1 + i)
    # HIP events around the scoring kernel (on the engine's own stream) of the timed steps
    eng.profile_kernel(True, every=args.event_every)
    fence()
    t0 = time.perf_counter()
    for i in range(args.steps):
        res = one_step(args.seed + 1000 + i)
    fence()
    dt = time.perf_counter() - t0
    kern_ms_total, kern_n = eng.profile_read()
    eng.profile_kernel(False)
    # second figure: the same K steps on the patches now resident (no preparation kernels)
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_step()
    fence()
    dt_resident = time.perf_counter() - t0

    # Known-answer step, outside the timed region: heading a* looks at a view stored on the LAST rank, so the global
    # decision has to come through the exchange (a stale or mis-ordered record gives the previous answer instead).
    from navsim_amd import synth
    a_star, f_star = A // 2, (world - 1) * F + (12345 % F)
    probe = synth.synth_patches(args.seed, A, h, w, full_range_s=args.full_range_s)
    probe[a_star] = synth.synth_views(args.seed, 1, h, w, first_view=f_star, full_range_s=args.full_range_s)[0]
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

    rccl_ranks, exchange_us = None, None
    if exchange is not None and hasattr(exchange, "rccl_ranks"):
        try:
            rccl_ranks = exchange.rccl_ranks()                      # ncclCommCount of the communicator the per-step collective uses
            exchange_us = exchange.exchange_only_us(50)             # every rank takes part: a collective
        except Exception as e:                                      # noqa: BLE001
            sys.stderr.write("rank %d: exchange facts unavailable: %r\n" % (rank, e))
    if use_dist:
        t = torch.tensor([dt, dt_resident], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt, dt_resident = float(t[0].item()), float(t[1].item())

    if rank == 0:
        comparisons = float(world) * F * A * args.steps
        kern_ms = kern_ms_total / max(kern_n, 1)
        shape = eng.workgroup_shape(A)
        workload = workload_name(w, h, F, A, cw, world) + (", saturation 0..127 (mixed layout)" if args.full_range_s else "")
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
                "timed_region": "per step: fresh patches made in HBM with their coefficient images and constants "
                                "(k_patch_prep, generator mode) + scoring kernel + reductions + decision + result read-back.  The "
                                "sensor model proper (k_patch_prep in sensing mode: same kernel, landscape fetches instead "
                                "of the generator) is timed in the `agent` block; the per-view scene_familiarity output "
                                "(plot-only in the reference, NavBySceneFamiliarity.py:301-303) is not produced",
                "precision": "the reference's arithmetic: uint8 HSV in, exact integer sums, float64 scores "
                             "(util.pyx:31-73); BASELINE.json's 'fp32' SSD wording is the ssd_f32 block",
                "scoring_kernel": kernel_of_shape(shape), "workgroup_shape": shape,
                "bytes_per_pixel_reference": 3 if cw > 0 else 1,
                "bytes_per_pixel_streamed": ((info["code_tile_bytes"] or info["bit_tile_bytes"]) / float(((F + 63) // 64) * 64 * h * w)
                                             if shape == 6 else info["n_planes"]),
                "parallelism": "library sharded x%d" % world,
                "rccl_ranks": rccl_ranks, "exchange_us_per_step": exchange_us,
                "exchange_words": (A + 4 * world) if use_dist else 0,
                "exchange": "none" if not use_dist else "1 exchange of per-heading records per step (%s)" % (
                    "mailbox in host-shared memory, no collective" if args.exchange == "mailbox" else
                    (("RCCL on the step's stream, device-resident"
                      if (exchange is not None and exchange.direct is not None) else "RCCL via torch.distributed, device-resident")
                     if args.backend == "nccl" else args.backend)),
            },
            "nav_steps_per_s": args.steps / dt,
            "scoring_only": {"value": comparisons / dt_resident, "ms_per_step": dt_resident / args.steps * 1e3,
                             "what": "the same K steps on resident patches (no k_patch_prep): the figure "
                                     "rounds 1-2 reported as the headline"},
            "layout": {
                "chosen": ("thermometer bit planes on the matrix cores" if shape == 6 else "byte planes, v_sad_u8"),
                "why": ("every stored byte plane takes few values (%d + %d planes per pixel = %.3g B/px instead of the "
                        "reference's %d): data-dependent -- a library with many saturation or value levels keeps byte "
                        "planes" % (info["bit_planes_hs"], info["bit_planes_v"], (info["bit_planes_hs"] + info["bit_planes_v"]) / 8.0,
                                    3 if cw > 0 else 1)) if shape == 6 else
                       "the library's byte planes take too many values for thermometer bit planes (or they were slower when timed)",
                "frac_algorithmic_is": "SURVEY 8(d) reference bytes (3 B/px) / kernel time / peak: above 1 because the layout is a "
                                       "lossless re-coding of this library's few levels, a compression ratio and not a bandwidth",
            },
            "best_heading": int(res["best_idex"]),
            "known_answer_step": "SKIPPED" if args.skip_known_answer else "ok on every rank (heading %d, view %d of %d, through the exchange)" % (a_star, f_star, world * F),
            "roofline": roofline_block(eng, info, shape, kern_ms, kern_n, F, h, w, A, cw, workload),
        }
        extras = world == 1 and not args.force_dist
        eng.clear_library()                                      # make room for the secondary blocks
        if extras and args.secondary:
            try:
                out["configs1"] = secondary_scoring(device_index, args.seed, 50000, 64, 64, 16, cw, 200, 20)
            except Exception as e:                               # noqa: BLE001 - an extra block must not cost the JSON line
                out["configs1"] = {"error": repr(e)}
        if extras and args.secondary and not args.full_range_s:
            try:
                out["full_range_s"] = secondary_scoring(device_index, args.seed, F, h, w, A, cw, max(args.steps // 5, 5), 3, full_range_s=True)
            except Exception as e:                               # noqa: BLE001
                out["full_range_s"] = {"error": repr(e)}
        if extras and args.agent_steps > 0:
            try:
                rates, n_lib, (ens_rate, ens_rate_real), outliers = agent_steps_per_s(64, 64, 16, cw, 50000, args.seed, args.agent_steps)
                out["agent"] = {"nav_steps_per_s": rates.get(False), "nav_steps_per_s_fake": rates.get(True),
                                "view_comparisons_per_s": (rates.get(False) or 0.0) * n_lib * 16, "library_views": n_lib,
                                "ensemble_of_32_nav_steps_per_s": ens_rate,
                                "ensemble_of_32_nav_steps_per_s_with_metrics": ens_rate_real,
                                "median_step_us_and_steps_over_4x_median": {"fake": outliers.get(True), "not_fake": outliers.get(False)},
                                "what": "navsim_amd.NavBySceneFamiliarity.step_forward() on the configs[1] shape (64x64, 16 "
                                        "headings, 50 000-view training path): sensor model on the GPU (2000x2000 landscape "
                                        "resident), scoring, decision, position update and the error metrics of "
                                        "NavBySceneFamiliarity.py:252-276; `_fake` = step_forward(fake=True), which skips "
                                        "the metrics as the reference's own fake flag does; Python caller"}
            except Exception as e:                               # noqa: BLE001
                out["agent"] = {"error": repr(e)}
        if extras and args.secondary:
            try:
                out["ssd_f32"] = ssd_f32_block(device_index, 50000, 64, 64, 16, 50)
            except Exception as e:                               # noqa: BLE001
                out["ssd_f32"] = {"error": repr(e)}
            try:                                                 # configs[2] as BASELINE.json words it: fp32, 32.8 GB
                out["ssd_f32_configs2"] = ssd_f32_block(device_index, F, h, w, A, 12)
            except Exception as e:                               # noqa: BLE001
                out["ssd_f32_configs2"] = {"error": repr(e)}
            try:                                                 # the headline library on the byte path (what a library the planes cannot describe pays)
                out["byte_path"] = secondary_scoring(device_index, args.seed, F, h, w, A, cw, max(args.steps // 5, 5), 3, env={"DEJAVU_BITS": "0"})
            except Exception as e:                               # noqa: BLE001
                out["byte_path"] = {"error": repr(e)}
            try:
                out["generic_hue"] = generic_block(device_index, args.seed, 50000, 64, 64, 16, cw, 50)
            except Exception as e:                               # noqa: BLE001
                out["generic_hue"] = {"error": repr(e)}
        if extras and args.secondary:
            try:
                out["ssd_u8"] = ssd_u8_block(device_index, 50000, 64, 64, 16, 50)
            except Exception as e:                               # noqa: BLE001
                out["ssd_u8"] = {"error": repr(e)}
        if extras and args.batch_agents > 0:
            try:
                out["ensemble"] = ensemble_comparisons_per_s(64, 64, 16, cw, args.seed, args.batch_agents, 100000, 5)
            except Exception as e:                               # noqa: BLE001 - an extra block must not cost the JSON line
                out["ensemble"] = {"error": repr(e)}
        if world == 1 and args.cpu_views > 0:
            out["cpu_baseline"] = cpu_baseline(h, w, A, cw, args.seed, min(args.cpu_views, F))
        else:
            out["cpu_baseline"] = None
        # The figures of the secondary blocks as scalars INSIDE `roofline` (the driver's record keeps config / roofline /
        # cpu_baseline whole and only lists other keys): every number DESIGN.md section 4 quotes can be read from here.
        def g(block, *path):
            v = out.get(block)
            for k in path:
                v = v.get(k) if isinstance(v, dict) else None
            return v
        def us(x):
            return None if x is None else x * 1e3
        out["roofline"]["secondary"] = {
            "c2_step_ms": out["ms_per_step"], "c2_kernel_ms": g("roofline", "kernel_ms"), "c2_scoring_only_ms": g("scoring_only", "ms_per_step"),
            "c1_step_us": us(g("configs1", "ms_per_step")), "c1_scoring_only_step_us": us(g("configs1", "scoring_only", "ms_per_step")),
            "c1_kernel_us": us(g("configs1", "roofline", "kernel_ms")), "c1_frac": g("configs1", "roofline", "frac"),
            "c1_view_comparisons_per_s": g("configs1", "value"),
            "full_s_ms": g("full_range_s", "ms_per_step"), "full_s_kernel_ms": g("full_range_s", "roofline", "kernel_ms"),
            "full_s_frac": g("full_range_s", "roofline", "frac"), "full_s_valu_frac": g("full_range_s", "roofline", "valu", "frac"),
            "byte_path_ms": g("byte_path", "ms_per_step"), "byte_path_kernel_ms": g("byte_path", "roofline", "kernel_ms"),
            "byte_path_frac": g("byte_path", "roofline", "frac"), "byte_path_valu_frac": g("byte_path", "roofline", "valu", "frac"),
            "generic_hue_step_us": us(g("generic_hue", "ms_per_step")), "generic_hue_kernel_us": us(g("generic_hue", "roofline", "kernel_ms")),
            "generic_hue_frac": g("generic_hue", "roofline", "frac"), "generic_hue_valu_frac": g("generic_hue", "roofline", "valu", "frac"),
            "ssd_f32_kernel_us": us(g("ssd_f32", "roofline", "kernel_ms")), "ssd_f32_frac": g("ssd_f32", "roofline", "frac"),
            "ssd_f32_step_us": us(g("ssd_f32", "ms_per_step")), "ssd_f32_direct_kernel_us": us(g("ssd_f32", "direct_form", "kernel_ms")),
            "ssd_f32_c2_kernel_ms": g("ssd_f32_configs2", "roofline", "kernel_ms"), "ssd_f32_c2_frac": g("ssd_f32_configs2", "roofline", "frac"),
            "ssd_f32_c2_step_ms": g("ssd_f32_configs2", "ms_per_step"), "ssd_f32_c2_direct_kernel_ms": g("ssd_f32_configs2", "direct_form", "kernel_ms"),
            "ssd_f32_c2_mfma_frac": g("ssd_f32_configs2", "roofline", "mfma", "frac"),
            "ssd_u8_kernel_us": us(g("ssd_u8", "roofline", "kernel_ms")), "ssd_u8_frac": g("ssd_u8", "roofline", "frac"),
            "ssd_u8_step_us": us(g("ssd_u8", "ms_per_step")),
            "ens_ms": g("ensemble", "ms_per_ensemble_step"), "ens_uploaded_ms": g("ensemble", "uploaded", "ms_per_ensemble_step"),
            "ens_mfma_frac": g("ensemble", "mfma_frac_of_peak"), "ens_mfma_form": g("ensemble", "mfma_form"),
            "ens_view_comparisons_per_s": g("ensemble", "view_comparisons_per_s"),
            "agent_steps_per_s": g("agent", "nav_steps_per_s"), "agent_steps_per_s_fake": g("agent", "nav_steps_per_s_fake"),
            "agent_median_step_us": g("agent", "median_step_us_and_steps_over_4x_median", "not_fake", "median_step_us"),
            "agent_ensemble_of_32_steps_per_s": g("agent", "ensemble_of_32_nav_steps_per_s"),
            "agent_ensemble_of_32_steps_per_s_with_metrics": g("agent", "ensemble_of_32_nav_steps_per_s_with_metrics"),
            "cpu_one_core_cmp_per_s": g("cpu_baseline", "value"), "cpu_16_threads_cmp_per_s": g("cpu_baseline", "multicore", "value"),
        }
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if exchange is not None:
        exchange.close()
    eng.close()
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
