"""Experiment loop around the agent: what one trial of the reference's farm runs and reports.

Counterpart of scripts/run_experiment.py:235-258 (`run_experiment`), :107-124 (`chop_path_to_len`) and the result
rows of its per-rank CSV files (:44-71 column formats, :320-343 header and row); the MPI farm, PNG loading and grain
labelling around them are out of scope.  Result keys, stop codes and the CSV text are the reference's: every function
here is pinned by outputs of the reference's own code (tests/golden/t7_experiment.npz, manifest.json "t7_experiment").
"""
import numpy as np

from .agent import StopNavigationException
from .synth import sin_training_path  # noqa: F401  (scripts/run_experiment.py:95-105; re-exported)

# ---- the CSV wire format of the farm (scripts/run_experiment.py:44-71): column -> format of its value ----------------
FLOAT_FORMAT = "{:6f}"
RESULT_FORMATS = {
    "path_coverage": FLOAT_FORMAT, "rmsd_error": FLOAT_FORMAT, "completed_frames": "{:d}", "stop_status": "{:d}",
    "n_captures": "{:d}", "percent_forgiving": FLOAT_FORMAT,
}
VARIABLE_FORMATS = {
    "landscape_class": "{}", "landscape_name": "{}", "training_path_curve": "{:4f}", "landscape_noise_factor": "{:4f}",
    "n_chemicals": "{:d}", "min_chem_grain_diameter": "{:4f}", "chem_weight": "{:4f}",
    "sensor_dimensions": "{0[0]:d};{0[1]:d};{0[2]:d};{0[3]:d}", "mask_middle_n": "{:d}", "n_sensor_levels": "{:d}",
    "step_size": "{:4f}", "saccade_degrees": "{:4f}", "n_test_angles": "{:d}", "start_offset": "{0[0]:4f};{0[1]:4f}",
    "landscape_flip_vertical": "{:d}", "landscape_flip_horizontal": "{:d}",      # booleans as ints
}


def csv_header(variables):
    """Header line of a task file: the trial variables then the result columns, each group sorted (:286-289, :320-324)."""
    return ", ".join(sorted(variables) + sorted(RESULT_FORMATS))


def csv_row(trial, result):
    """One result row: `trial` maps variable -> value, `result` is run_experiment's dict (:339-343)."""
    variables = sorted(trial)
    return (", ".join(VARIABLE_FORMATS[v].format(trial[v]) for v in variables) + ", " +
            ", ".join(RESULT_FORMATS[v].format(result[v]) for v in sorted(RESULT_FORMATS)))


def write_task_csv(path, rows):
    """A task file like the farm's task-<rank>.csv: `rows` = iterable of (trial, result); line-buffered like :318."""
    rows = list(rows)
    with open(path, "w", buffering=1) as f:
        if rows:
            print(csv_header(rows[0][0]), file=f)
        for trial, result in rows:
            print(csv_row(trial, result), file=f)

FRAME_FACTOR = 3.0               # frames = FRAME_FACTOR * training_path_length / step_size (scripts/run_experiment.py:21,238)
N_CONSECUTIVE_SCENES = 0.05      # window of the forgiving coverage / capture metrics (:28)


def chop_path_to_len(path, length):
    """Trim a polyline to at most `length`, dropping points from the front and the back alternately (:107-124)."""
    seg = np.linalg.norm(path[1:] - path[:-1], axis=1)
    assert np.sum(seg) >= length
    lo, hi = 0, len(path)
    drop_front = True
    while np.sum(seg[lo:hi]) > length and hi - lo > 0:
        if drop_front:
            lo += 1
        else:
            hi -= 1
        drop_front = not drop_front
    return path[lo:hi]


def run_experiment(nsf, frames=None):
    """Step the agent until it stops or `frames` run out; returns the reference's result row (:251-258)."""
    if frames is None:
        frames = int(FRAME_FACTOR * nsf.training_path_length / nsf.step_size)
    status = 0
    completed = 0
    try:
        for _ in range(frames):
            nsf.step_forward()
            completed += 1
    except StopNavigationException as stop:
        status = stop.get_code()
        nsf.stopped_with_exception = stop
    return {
        "path_coverage": nsf.percent_recapitulated,
        "rmsd_error": nsf.navigation_error,
        "completed_frames": completed,
        "stop_status": status,
        "percent_forgiving": nsf.percent_recapitulated_forgiving(n_consecutive_scenes=N_CONSECUTIVE_SCENES),
        "n_captures": nsf.n_captures(n_consecutive_scenes=N_CONSECUTIVE_SCENES),
    }


def run_ensemble(ens, frames=None):
    """run_experiment for a navsim_amd.NavEnsemble: every agent gets the reference's result row (:251-258)."""
    first = ens.agents[0]
    if frames is None:
        frames = int(FRAME_FACTOR * first.training_path_length / first.step_size)
    done = ens.run(frames)
    rows = []
    for i, nsf in enumerate(ens.agents):
        scored = nsf._n_navigation_error > 0                      # an agent stopped before its first step has no error yet
        rows.append({
            "path_coverage": nsf.percent_recapitulated,
            "rmsd_error": nsf.navigation_error if scored else float("nan"),
            "completed_frames": done[i],
            "stop_status": ens.stop_status[i],
            "percent_forgiving": nsf.percent_recapitulated_forgiving(n_consecutive_scenes=N_CONSECUTIVE_SCENES),
            "n_captures": nsf.n_captures(n_consecutive_scenes=N_CONSECUTIVE_SCENES),
        })
    return rows
