#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by RUNNING THE REFERENCE'S OWN CODE.

Runs only in the build container (needs /root/reference, Cython, gcc); the GPU box never
runs this and never sees the reference.  Nothing from the reference is written into the repo:
the reference package is compiled in a scratch directory under /tmp and imported from there.
What gets committed is data only: seeds/parameters, SHA-256 of the regenerated inputs, and the
reference's outputs.

    python3 tests/golden/make_golden.py [--reference /root/reference] [--out tests/golden]

How the reference is made importable here:
  * navsim/util.pyx does not compile under Cython >= 3.1 because `np.int_t` left Cython's
    numpy.pxd; the three occurrences (util.pyx:77,82,97 -- set_HS_where_equal and the
    downscale_chem histogram, none on the scored path) are rewritten to `np.int64_t` (what
    `np.int_t` meant on 64-bit Linux) while copying to the scratch dir.  util.pyx:28-73 (the
    kernel that is being pinned) is compiled byte-for-byte as it stands.
  * navsim/NavBySceneFamiliarity.py imports `skimage` at module level but never calls it, and
    uses the NumPy aliases `np.float`, `np.int`, `np.product` that NumPy 2 removed.  Empty
    `skimage`/`skimage.transform` modules and the three aliases are registered before import;
    no reference source line is edited.
"""
import argparse
import hashlib
import json
import os
import shutil
import subprocess
import sys
import tempfile
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(REPO, "navigation-by-deja-vu_amd"))

from navsim_amd import synth  # noqa: E402  (the build's own seeded input generator)


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def build_reference(ref_root):
    work = tempfile.mkdtemp(prefix="navsim_ref_", dir="/tmp")
    pkg = os.path.join(work, "navsim")
    os.makedirs(pkg)
    for name in os.listdir(os.path.join(ref_root, "navsim")):
        src = os.path.join(ref_root, "navsim", name)
        if name.endswith(".py"):
            shutil.copy(src, os.path.join(pkg, name))
        elif name == "util.pyx":
            with open(src) as f:
                text = f.read()
            with open(os.path.join(pkg, name), "w") as f:
                f.write(text.replace("np.int_t", "np.int64_t"))
    with open(os.path.join(work, "setup.py"), "w") as f:
        f.write(
            "from setuptools import setup, Extension\n"
            "from Cython.Build import cythonize\n"
            "import numpy as np\n"
            "setup(name='navsim_ref', ext_modules=cythonize([Extension('navsim.util',"
            " ['navsim/util.pyx'], include_dirs=[np.get_include()])]))\n")
    subprocess.run([sys.executable, "setup.py", "-q", "build_ext", "--inplace"], cwd=work, check=True,
                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return work


def import_reference(work):
    os.environ.setdefault("MPLBACKEND", "Agg")
    for m in ("skimage", "skimage.transform"):
        sys.modules.setdefault(m, types.ModuleType(m))
    for name, val in (("float", float), ("int", int), ("product", np.prod)):
        if not hasattr(np, name):
            setattr(np, name, val)
    sys.path.insert(0, work)
    import navsim  # noqa: F401
    import navsim.util
    return navsim


# ----------------------------------------------------------------------------------------------
def kernel_vectors(navsim, out):
    """T1: sads_hsv_metric (util.pyx:31-73) through the reference's own factory."""
    cases = []
    arrays = {}
    for (F, h, w) in ((64, 8, 8), (500, 32, 32), (130, 5, 7)):
        for kind in ("levels", "random"):
            seed = 1000 + F + h
            if kind == "levels":
                lib = synth.synth_views(seed, F, h, w)
                scene = synth.synth_patches(seed, 1, h, w)[0]
            else:
                lib = synth.random_hsv(seed, (F, h, w, 3))
                scene = synth.random_hsv(seed + 1, (h, w, 3))
                # make hue collisions likely on full-range data too
                lib[..., 0] &= 0x03
                scene[..., 0] &= 0x03
            for cw in (0.0, 0.25, 0.3, 1.0):
                fam = np.full(F, np.nan)
                func = navsim.util.sads_familiarity(cw)(lib)
                func(scene, fam)
                key = "k_%d_%d_%d_%s_%g" % (F, h, w, kind, cw)
                arrays[key] = fam
                cases.append(dict(key=key, F=F, h=h, w=w, kind=kind, seed=seed, chem_weight=cw,
                                  lib_sha=sha(lib), scene_sha=sha(scene),
                                  max_familiarity=int(func.max_familiarity)))
    np.savez_compressed(os.path.join(out, "t1_kernel.npz"), **arrays)
    return cases


def run_step_with_patches(navsim, lib, patches, cw):
    """Drive the reference's step_forward (NavBySceneFamiliarity.py:279-329) on given patches.

    The instance's get_sensor_mat is replaced by a feeder so that the heading loop, the min-merge,
    np.max and np.argmax that run are the reference's own lines.
    """
    F, h, w, _ = lib.shape
    A = patches.shape[0]
    land = np.zeros((8, 8, 3), dtype=np.uint8)
    nsf = navsim.NavBySceneFamiliarity(land, (w, h), 1.0, n_test_angles=A,
                                       familiarity_model=navsim.util.sads_familiarity(cw))
    nsf.familiar_scenes = lib
    nsf.scene_familiarity = np.zeros(F)
    nsf.training_path = np.zeros((F, 2))
    nsf._familiarity_func = nsf.familiarity_model(lib)
    feed = iter(patches)
    nsf.get_sensor_mat = lambda position, angle: next(feed)
    angle0 = 0.3
    nsf.angle = angle0
    nsf.step_forward(fake=True)
    best = int(np.argmax(nsf.angle_familiarity))
    # cross-check: the heading actually taken is the one argmax reports
    assert np.isclose(nsf.angle, (angle0 + nsf.angle_offsets[best]) % (2 * np.pi))
    return nsf.angle_familiarity.copy(), nsf.scene_familiarity.copy(), best, float(nsf.step_familiarity)


def step_vectors(navsim, out):
    """T2 (plain) and T3 (tie stress) step fixtures."""
    cases = []
    arrays = {}
    specs = [
        # name, F, h, w, A, cw, kind
        ("s_c0", 500, 32, 32, 8, 0.0, "levels"),
        ("s_c0_cw", 500, 32, 32, 8, 0.25, "levels"),
        ("s_near", 300, 16, 16, 10, 0.5, "near"),
        ("s_rand", 200, 12, 20, 5, 0.3, "random"),
        ("s_ties_8x8", 20000, 8, 8, 8, 0.0, "levels"),
        ("s_ties_8x8_cw", 6000, 8, 8, 8, 0.5, "levels"),
        ("s_ties_4x4", 5000, 4, 4, 16, 0.0, "levels"),
        ("s_dup", 256, 8, 8, 6, 0.0, "dup"),
    ]
    for name, F, h, w, A, cw, kind in specs:
        seed = 7000 + len(cases)
        if kind == "random":
            lib = synth.random_hsv(seed, (F, h, w, 3))
            lib[..., 0] &= 0x07
            patches = synth.random_hsv(seed + 1, (A, h, w, 3))
            patches[..., 0] &= 0x07
        else:
            lib = synth.synth_views(seed, F, h, w)
            patches = synth.synth_patches(seed, A, h, w)
            if kind == "near":
                patches[3] = synth.near_match_patch(lib[F // 3], seed)
                patches[7] = synth.near_match_patch(lib[F // 2], seed + 9)
            if kind == "dup":
                # every view identical to every patch except a few pixels: massive exact ties
                lib[:] = lib[0]
                patches[:] = lib[0]
                patches[:, 0, 0, 2] = 255 - lib[0, 0, 0, 2]
                lib[7, 1, 1, 2] ^= 0xFF
        ang, scn, best, stepfam = run_step_with_patches(navsim, lib, patches, cw)
        arrays[name + "_angle"] = ang
        arrays[name + "_scene"] = scn
        # reference's own per-heading argmax over the library (np.argmax on its fambuf)
        func = navsim.util.sads_familiarity(cw)(lib)
        tmp = np.empty(F)
        func(patches[best], tmp)
        best_view = int(np.argmax(tmp))
        cases.append(dict(name=name, F=F, h=h, w=w, A=A, chem_weight=cw, kind=kind, seed=seed,
                          lib_sha=sha(lib), patches_sha=sha(patches), best_idex=best,
                          best_view=best_view, step_familiarity=stepfam))
    np.savez_compressed(os.path.join(out, "t2_step.npz"), **arrays)
    return cases


def trajectory(navsim, out):
    """T4: full step_forward trajectories on a synthetic landscape (fake=False)."""
    cases = []
    arrays = {}
    specs = [
        # name, cw, sensor(w,h), pixel dims, step, A, levels, mask, saccade, n_views, n_steps
        ("traj_c0", 0.0, (32, 32), [1, 1], 0.5, 8, 5, 0, 180.0, 560, 1000),
        ("traj_cw", 0.5, (32, 32), [1, 1], 0.5, 8, 5, 0, 180.0, 560, 1000),
        ("traj_px", 0.25, (16, 8), [2, 4], 1.0, 10, 4, 1, 90.0, 300, 250),
    ]
    L, grain, lseed = 900, 4, 424242
    land = synth.synth_landscape(lseed, L, grain)
    for name, cw, sdim, spd, step, A, levels, mask, sacc, n_views, n_steps in specs:
        path = synth.sin_training_path(0.5, 0.2 * L, 0.6 * L, arclen=1.0)[:n_views]
        nsf = navsim.NavBySceneFamiliarity(
            land, sdim, step, n_test_angles=A, sensor_pixel_dimensions=spd,
            n_sensor_levels=levels, mask_middle_n=mask, saccade_degrees=sacc,
            max_distance_to_training_path=450,
            familiarity_model=navsim.util.sads_familiarity(cw))
        nsf.train_from_path(path)
        d = path[2] - path[1]
        nsf.angle = float(np.arctan2(d[1], d[0]) % (2 * np.pi)) + np.deg2rad(7.0)
        nsf.position = path[1] + np.array([1.5, -1.0])
        best, pos, ang, fam = [], [], [], []
        status = 0
        try:
            for _ in range(n_steps):
                nsf.step_forward()
                best.append(int(np.argmax(nsf.angle_familiarity)))
                pos.append([float(nsf.position[0]), float(nsf.position[1])])
                ang.append(float(nsf.angle))
                fam.append(float(nsf.step_familiarity))
        except navsim.StopNavigationException as e:
            status = e.get_code()
            # ReachedEnd/TooFar are raised after the move (NavBySceneFamiliarity.py:322-329);
            # OutOfBounds is raised mid-loop before any move (:156-158) and records nothing.
            if not isinstance(e, navsim.OutOfLandscapeBoundsException):
                best.append(int(np.argmax(nsf.angle_familiarity)))
                pos.append([float(nsf.position[0]), float(nsf.position[1])])
                ang.append(float(nsf.angle))
                fam.append(float(nsf.step_familiarity))
        arrays[name + "_best"] = np.asarray(best, dtype=np.int32)
        arrays[name + "_pos"] = np.asarray(pos)
        arrays[name + "_angle"] = np.asarray(ang)
        arrays[name + "_fam"] = np.asarray(fam)
        arrays[name + "_scenes_sha"] = np.frombuffer(bytes.fromhex(sha(nsf.familiar_scenes)), dtype=np.uint8)
        arrays[name + "_last_scene_fam"] = nsf.scene_familiarity.copy()
        arrays[name + "_last_angle_fam"] = nsf.angle_familiarity.copy()
        cases.append(dict(
            name=name, chem_weight=cw, sensor_dimensions=list(sdim), sensor_pixel_dimensions=spd,
            step_size=step, n_test_angles=A, n_sensor_levels=levels, mask_middle_n=mask,
            saccade_degrees=sacc, n_views=int(len(path)), n_steps=n_steps,
            landscape=dict(seed=lseed, size=L, grain=grain, sha=sha(land)),
            start_angle_offset_deg=7.0, start_offset=[1.5, -1.0],
            steps_recorded=len(best), stop_status=status,
            navigated_for_frames=int(nsf.navigated_for_frames),
            navigation_error=float(nsf.navigation_error),
            percent_recapitulated=float(nsf.percent_recapitulated),
            percent_forgiving=float(nsf.percent_recapitulated_forgiving(0.05)),
            n_captures=int(nsf.n_captures(0.05)),
            training_path_length=float(nsf.training_path_length),
            path_sha=sha(path)))
    np.savez_compressed(os.path.join(out, "t4_trajectory.npz"), **arrays)
    return cases


def sensor_vectors(navsim, out):
    """T5: get_sensor_mat / fill_sensor_from / downscale_chem outputs incl. their quirks."""
    cases = []
    arrays = {}
    L, grain, lseed = 300, 3, 99
    land = synth.synth_landscape(lseed, L, grain)
    # richer S and V so that the block statistics of downscale_chem are exercised
    extra = synth.random_hsv(5, (L, L, 3))
    land2 = land.copy()
    land2[..., 1] = np.where(land[..., 1] > 0, extra[..., 1], 0)
    land2[..., 2] = extra[..., 2]
    land2[..., 0] = (extra[..., 0] % 3) * 85
    arrays["land2"] = land2
    specs = [
        ("g11", land, (32, 32), [1, 1], 5, 0),
        ("g11m", land, (20, 12), [1, 1], (256, 256, 3), 3),
        ("g24", land2, (16, 8), [2, 4], 4, 1),
        ("g42", land2, (10, 6), [4, 2], (7, 256, 256), 0),
        ("g22", land2, (8, 8), [2, 2], 6, 0),
    ]
    poses = [(150.0, 150.0, 0.0), (100.3, 170.8, np.pi / 2), (171.49, 99.5, 1.2345),
             (80.5, 80.5, 4.0), (200.25, 120.75, 2 * np.pi - 0.01), (149.5, 150.5, np.pi)]
    for name, lnd, sdim, spd, levels, mask in specs:
        nsf = navsim.NavBySceneFamiliarity(lnd, sdim, 1.0, n_test_angles=4,
                                           sensor_pixel_dimensions=spd, n_sensor_levels=levels,
                                           mask_middle_n=mask)
        mats = []
        glimpses = []
        for (x, y, a) in poses:
            mats.append(nsf.get_sensor_mat((x, y), a).copy())
            glimpses.append(nsf._landscape_glimpse_buf.copy())
        arrays[name + "_mats"] = np.stack(mats)
        arrays[name + "_glimpses"] = np.stack(glimpses)
        cases.append(dict(name=name, landscape="land2" if lnd is land2 else "land",
                          sensor_dimensions=list(sdim), sensor_pixel_dimensions=spd,
                          n_sensor_levels=list(levels) if isinstance(levels, tuple) else levels,
                          mask_middle_n=mask, poses=[list(p) for p in poses]))
    # downscale_chem on its own, including the integer-division quirk (util.pyx:131)
    blk = np.full((8, 2, 3), 255, dtype=np.uint8)
    arrays["dc_quirk_in"] = blk
    arrays["dc_quirk_out"] = navsim.util.downscale_chem(blk, 8, 2)
    meta = dict(landscape=dict(seed=lseed, size=L, grain=grain, sha=sha(land)), land2_sha=sha(land2))
    np.savez_compressed(os.path.join(out, "t5_sensor.npz"), **arrays)
    return dict(meta=meta, cases=cases)


def ssds_vectors(navsim, out):
    rng_bytes = synth.random_hsv(31337, (2, 37, 53))
    a = rng_bytes[0].astype(np.float64) / 7.0
    b = rng_bytes[1].astype(np.float64) / 3.0
    val = float(navsim.util.ssds(a, b))
    np.savez_compressed(os.path.join(out, "t6_ssds.npz"), a=a, b=b, ssd=np.array(val))
    return dict(ssd=val)


def import_run_experiment(ref_root):
    """scripts/run_experiment.py imported as a module (its __main__ block does not run).  It needs, at import time only,
    mpi4py and skimage.measure / skimage.filters.rank, which this container lacks: empty stand-in MODULES are registered
    for the import statements (nothing of them is ever called by the functions recorded here -- sin_training_path,
    chop_path_to_len, run_experiment and the two format tables are plain Python/NumPy), and the NumPy-2-removed
    aliases its navsim.generate_landscapes import touches are already in place (import_reference)."""
    import importlib.util
    for m in ("mpi4py", "skimage.measure", "skimage.filters", "skimage.filters.rank"):
        sys.modules.setdefault(m, types.ModuleType(m))
    sys.modules["mpi4py"].MPI = types.SimpleNamespace()
    sys.modules["skimage"].measure = sys.modules["skimage.measure"]
    sys.modules["skimage"].filters = sys.modules["skimage.filters"]
    sys.modules["skimage.filters"].rank = sys.modules["skimage.filters.rank"]
    spec = importlib.util.spec_from_file_location("ref_run_experiment", os.path.join(ref_root, "scripts", "run_experiment.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def experiment_vectors(navsim, ref_root, out):
    """T7: the experiment helpers of scripts/run_experiment.py -- sin_training_path (:95-105), chop_path_to_len
    (:107-124), run_experiment (:235-258) and the CSV row of the farm (:44-71, :339-343)."""
    rx = import_run_experiment(ref_root)
    arrays = {}
    paths = []
    for k, (curve, start, length, arclen) in enumerate(((0.5, 180.0, 540.0, 1.0), (0.0, 40.0, 120.0, 0.35),
                                                        (1.0, 400.0, 1200.0, 2.0), (0.25, 10.5, 77.25, 0.1))):
        arrays["sin_%d" % k] = rx.sin_training_path(curve, start, length, arclen=arclen)
        paths.append(dict(key="sin_%d" % k, curve=curve, start_x=start, l=length, arclen=arclen))
    chops = []
    for k, (src, frac) in enumerate((("sin_0", 0.5), ("sin_1", 0.999), ("sin_2", 0.123), ("sin_3", 1.0))):
        p = arrays[src]
        total = float(np.sum(np.linalg.norm(p[1:] - p[:-1], axis=1)))
        arrays["chop_%d" % k] = rx.chop_path_to_len(p, frac * total)
        chops.append(dict(key="chop_%d" % k, path=src, length=frac * total))
    # one trial of the farm on the T4 landscape: the agent of traj_px, run by the reference's run_experiment
    L, grain, lseed = 900, 4, 424242
    land = synth.synth_landscape(lseed, L, grain)
    tp = synth.sin_training_path(0.5, 0.2 * L, 0.6 * L, arclen=1.0)[:300]
    rows = []
    for name, cw, frames in (("row_default_frames", 0.25, None), ("row_40_frames", 0.0, 40)):
        nsf = navsim.NavBySceneFamiliarity(land, (16, 8), 1.0, n_test_angles=10, sensor_pixel_dimensions=[2, 4],
                                           n_sensor_levels=4, mask_middle_n=1, saccade_degrees=90.0,
                                           max_distance_to_training_path=450,
                                           familiarity_model=navsim.util.sads_familiarity(cw))
        nsf.train_from_path(tp)
        d = tp[2] - tp[1]
        nsf.angle = float(np.arctan2(d[1], d[0]) % (2 * np.pi)) + np.deg2rad(7.0)
        nsf.position = tp[1] + np.array([1.5, -1.0])
        res = rx.run_experiment(nsf, frames=frames)
        trial = dict(landscape_class="synthetic", landscape_name="land_%d.png" % lseed, training_path_curve=0.5,
                     landscape_noise_factor=0.0, n_chemicals=2, min_chem_grain_diameter=2.0, chem_weight=cw,
                     sensor_dimensions=[16, 8, 2, 4], mask_middle_n=1, n_sensor_levels=4, step_size=1.0,
                     saccade_degrees=90.0, n_test_angles=10, start_offset=[0.25, 7.0], landscape_flip_vertical=0,
                     landscape_flip_horizontal=1)
        variables = sorted(trial)
        result_vars = sorted(rx.result_variables)
        header = ", ".join(variables + result_vars)
        line = ", ".join([rx.variable_formats[v].format(trial[v]) for v in variables]) + ", " + \
               ", ".join([rx.result_variables[v].format(res[v]) for v in result_vars])
        rows.append(dict(name=name, chem_weight=cw, frames=frames, trial=trial, header=header, line=line,
                         result={k: (int(v) if isinstance(v, (int, np.integer)) else float(v)) for k, v in res.items()}))
    np.savez_compressed(os.path.join(out, "t7_experiment.npz"), **arrays)
    return dict(paths=paths, chops=chops, rows=rows, frame_factor=float(rx.FRAME_FACTOR),
                n_consecutive_scenes=float(rx.N_CONSECUTIVE_SCENES),
                landscape=dict(seed=lseed, size=L, grain=grain, sha=sha(land)), n_views=300)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reference", default="/root/reference")
    ap.add_argument("--out", default=HERE)
    ap.add_argument("--only", default=None, help="regenerate one fixture group only (e.g. t7_experiment) into the existing manifest")
    args = ap.parse_args()
    work = build_reference(args.reference)
    try:
        navsim = import_reference(work)
        if args.only == "t7_experiment":
            with open(os.path.join(args.out, "manifest.json")) as f:
                manifest = json.load(f)
            manifest["t7_experiment"] = experiment_vectors(navsim, args.reference, args.out)
            with open(os.path.join(args.out, "manifest.json"), "w") as f:
                json.dump(manifest, f, indent=1, sort_keys=True)
            print("t7_experiment written to", args.out)
            return
        manifest = dict(
            generator="tests/golden/make_golden.py",
            numpy=np.__version__,
            note="outputs of the reference's own code (navsim util.pyx + NavBySceneFamiliarity.py) "
                 "on inputs regenerated from seeds by navsim_amd.synth",
            t1_kernel=kernel_vectors(navsim, args.out),
            t2_step=step_vectors(navsim, args.out),
            t4_trajectory=trajectory(navsim, args.out),
            t5_sensor=sensor_vectors(navsim, args.out),
            t6_ssds=ssds_vectors(navsim, args.out),
            t7_experiment=experiment_vectors(navsim, args.reference, args.out),
        )
        with open(os.path.join(args.out, "manifest.json"), "w") as f:
            json.dump(manifest, f, indent=1, sort_keys=True)
    finally:
        shutil.rmtree(work, ignore_errors=True)
    print("golden fixtures written to", args.out)


if __name__ == "__main__":
    main()
